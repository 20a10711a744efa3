"""CPU checks behind TRXSIG_SOFT_TOLERANCE (csrc/trxsig_demod.h, fused_demod_tol): the constants of its error bound hold for the
real tables, and the bound itself holds on seeded bursts -- the reference-order arithmetic (taken from the ORACLE's soft bits) and a
numpy emulation of the rearranged arithmetic (samples unscaled, fused multiply-adds, 1/amp folded into the reverse rotation) are both
measured against the float64 value of the same real number, and against each other.  No GPU: the HIP kernel itself is graded against
the oracle in tests/test_gpu_soft_tolerance.py."""
import numpy as np
import pytest

import _pkg
import oraclebind
import synth

U = 2.0 ** -24


@pytest.fixture(scope="module")
def pkg():
    return _pkg.load()


def tables(pkg, sps):
    return pkg.build_tables_host(sps).view(pkg.tables_dtype())[0]


@pytest.mark.parametrize("sps", [1, 2, 4])
def test_constants_of_the_bound(pkg, sps):
    T = tables(pkg, sps)
    S = np.abs(T["sinc_grid"][:, :21].astype(np.float64)).sum(axis=1)
    assert S.max() <= 4.0, S.max()                           # S <= 4 in the derivation
    assert not T["sinc_grid"][:, 21:24].any()               # (the row's padding never adds to a sum)
    rv = T["rev"][: 157 * sps : sps]
    assert (np.abs(rv.real) + np.abs(rv.imag)).max() <= 1.5  # |c| + |d| <= 1.5 (the table holds (+-1, eps) pairs: 1.0000001)


def f32(x):
    return np.asarray(x, np.float64).astype(np.float32)


def fma32(a, b, c):
    """fl32(a * b + c): the product of two float32 is exact in float64; the float64 sum is within 2^-53 relative of the real
    one -- a double rounding that can move a float32 result by one unit in 2^-29 of the cases, far below what is measured here."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


@pytest.mark.parametrize("sps,sigmas", [(4, (0.0, 0.1, 0.3)), (1, (0.0, 0.3)), (2, (0.1,))])
def test_bound_on_seeded_bursts(pkg, sps, sigmas):
    T = tables(pkg, sps)
    o = oraclebind.Oracle(sps)
    tsc, B = 3, 256
    x, off, length, _ = synth.normal_batch(sps, B, tsc, seed=4242 + sps, sigmas=sigmas)
    ok, amp, toa, soft = o.normal_batch(x, off, length, tsc, nsoft=148, nthreads=4)
    worst_ref = worst_tol = worst_pair = 0.0
    n_used = 0
    for b in np.flatnonzero(ok.astype(bool)):
        xs = x[off[b]:off[b] + length[b]]
        N = len(xs)
        a = np.complex64(amp[b])
        # ((complex)1.0)/channel exactly as Complex.h does it (float32, every operation rounded)
        ar, ai = np.float32(a.real), np.float32(a.imag)
        n2 = np.float32(np.float32(ai * ai) + np.float32(ar * ar))
        cr, ci = np.float32(ar / n2), np.float32(-ai / n2)
        inv_r = np.float32(np.float32(np.float32(1) * cr) - np.float32(np.float32(0) * ci))
        inv_i = np.float32(np.float32(np.float32(1) * ci) + np.float32(np.float32(0) * cr))
        delay = np.float32(-toa[b])
        io = int(np.floor(delay))
        frac = np.float32(delay - np.float32(io))
        f512 = np.float32(frac * np.float32(512))
        f = int(f512)
        assert f < 512 and np.float32(f) == f512                 # peakDetect leaves TOA on the 1/512 grid
        if not abs(float(frac)) > 1e-2:
            continue                                         # (no filter: a plain copy, nothing to bound)
        tp = T["sinc_grid"][f, :21]
        m = np.arange(148)
        t = sps * m - io
        valid = (t >= 0) & (t < N)
        # filtered[t] = sum_j tap[j] * scaled[t + 10 - j]  (convolve NO_DELAY, :590), zeros outside the burst
        idx = t[:, None] + 10 - np.arange(21)[None, :]
        inside = (idx >= 0) & (idx < N)
        xw = np.where(inside, xs[np.clip(idx, 0, N - 1)], 0).astype(np.complex64)        # [148, 21]
        rv = T["rev"][sps * m]
        inv64 = complex(float(inv_r), float(inv_i))
        R = ((rv.astype(np.complex128) * inv64) * (xw.astype(np.complex128) * tp.astype(np.float64)[None, :]).sum(axis=1)).real
        Z = max(np.abs(xs.real).max(), np.abs(xs.imag).max()) * (abs(float(inv_r)) + abs(float(inv_i)))
        if Z > 8.0:
            continue                                         # the kernel hands such a burst to the exact code
        n_used += 1
        # (1) the reference order, from the oracle's soft bits: re = 2 soft - 1 where the slicer did not clip
        sv = soft[b, :148].astype(np.float64)
        unclipped = valid & (sv > 0.0) & (sv < 1.0)
        re_ref = 2.0 * sv - 1.0                              # exact inverse up to the rounding of re + 1 (<= u)
        err_ref = np.abs(re_ref - R)[unclipped]
        S = float(np.abs(tp.astype(np.float64)).sum())
        cd = (np.abs(rv.real) + np.abs(rv.imag)).astype(np.float64)
        bound_ref = (26.2 * U * cd * S * Z + 2 * U)[unclipped]
        assert np.all(err_ref <= bound_ref), (b, (err_ref / bound_ref).max())
        worst_ref = max(worst_ref, (err_ref / (U * Z)).max() if err_ref.size else 0.0)
        # (2) the rearranged form, emulated
        yr = np.zeros(148, np.float32); yi = np.zeros(148, np.float32)
        for j in range(21):
            tj = np.full(148, tp[j], np.float32)
            yr = fma32(xw[:, j].real.astype(np.float32), tj, yr)
            yi = fma32(xw[:, j].imag.astype(np.float32), tj, yi)
        c, d = rv.real.astype(np.float32), rv.imag.astype(np.float32)
        ir = np.full(148, inv_r, np.float32); ii = np.full(148, inv_i, np.float32)
        aa = fma32(c, ir, -(d * ii).astype(np.float32))
        bb = fma32(c, ii, (d * ir).astype(np.float32))
        re_tol = fma32(aa, yr, -(bb * yi).astype(np.float32)).astype(np.float64)
        err_tol = np.abs(re_tol - R)[valid]
        assert np.all(err_tol <= (25.0 * U * cd * S * Z)[valid] + 1e-30), (b, err_tol.max())
        worst_tol = max(worst_tol, (err_tol / (U * Z)).max())
        pair = np.abs(re_tol - re_ref)[unclipped]
        assert np.all(pair <= 308.0 * U * Z + 2 * U)
        worst_pair = max(worst_pair, (pair / (U * Z)).max() if pair.size else 0.0)
    assert n_used >= B // 8, n_used
    # the typical error is far inside the bound: a few u * Z
    assert worst_pair < 40.0, (worst_ref, worst_tol, worst_pair)


def test_library_exports_the_soft_mode_switch(pkg):
    L = pkg.lib()
    assert hasattr(L, "trxsig_set_soft_mode") and hasattr(L, "trxsig_get_soft_mode")
    assert pkg.SOFT_EXACT == 0 and pkg.SOFT_TOLERANCE == 1
