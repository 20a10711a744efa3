"""GPU parity of the TOLERANCE-mode demodulator (trxsig_set_soft_mode(ctx, TRXSIG_SOFT_TOLERANCE); csrc/trxsig_demod.h,
fused_demod_tol) against the oracle and the golden vectors captured from the real reference:
  * flags, amplitude, TOA, avgPwr: value-exact (IEEE ==), as in the exact mode;
  * hard bits: identical, every burst, every bit;
  * soft bits: |soft - reference soft| <= 7.4e-5 on the [0, 1] scale (the guaranteed bound; north_star allows 1e-4), and the
    error actually measured is reported and held under 2e-6;
  * a burst the fast form must not take (NaN / infinity anywhere, max|x| * |1/amp| > 8, a TOA off the 1/512 grid, a soft symbol
    within the guard band of the slicer's 0.5) equals the exact mode bit pattern for bit pattern.
The exact mode stays the default and is what every other test file grades."""
import numpy as np
import pytest

import _pkg
import oraclebind
import synth
from util import GpuBatch, assert_veq

pytestmark = pytest.mark.gpu

BOUND = 7.4e-5          # guaranteed; the kernel hands over anything it cannot guarantee
MEASURED = 2e-6         # what the arithmetic actually does on these inputs (a few 2^-24 * Z)


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _pkg.load()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = {s: pkg.TrxSig(s, 0) for s in (1, 2, 4)}
    for v in c.values():
        v.use_torch_stream()
        v.set_soft_mode(pkg.SOFT_TOLERANCE)
        assert v.soft_mode() == pkg.SOFT_TOLERANCE
    return c


def grade(soft, ref, what):
    """soft bits of detected bursts against the reference's; returns (max abs error, fraction of values that are not identical)."""
    soft = np.asarray(soft, np.float64); ref = np.asarray(ref, np.float64)
    assert not np.isnan(soft).any() and not np.isnan(ref).any(), what
    err = np.abs(soft - ref)
    assert err.max(initial=0.0) <= BOUND, (what, err.max())
    return float(err.max(initial=0.0)), float((err > 0).mean()) if err.size else 0.0


@pytest.mark.parametrize("name", ["normal_sps4.npz", "normal_sps1.npz"])
def test_golden_normal_tolerance(pkg, ctx, golden, name):
    g = golden(name)
    sps = int(g["sps"]); t = ctx[sps]
    worst = 0.0
    for tsc in range(8):
        sel = np.flatnonzero(g["tsc"] == tsc)
        gb = GpuBatch(g["x"], g["off"][sel], g["len"][sel], nsoft=148, stride=160)
        t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, avgpwr=gb.pwr,
                              hard=gb.hard, detect_thresh=3.0, energy_thresh=-1.0, nsoft=148, soft_stride=160)
        r = gb.results()
        det = (r["flags"] & pkg.F_DETECT) != 0
        assert_veq(det, g["ok"][sel].astype(bool), "detect flags tsc %d" % tsc)
        assert_veq(r["amp"], g["amp"][sel], "amp"); assert_veq(r["toa"], g["toa"][sel], "toa")
        assert_veq(r["pwr"], g["energy_pwr"][sel], "energyDetect avgPwr")
        for j, i in enumerate(sel):
            if det[j]:
                assert_veq(r["hard"][j, :148], (g["soft"][i, :148] > 0.5).astype(np.uint8), "hard %d" % i)
                e, _ = grade(r["soft"][j, :148], g["soft"][i, :148], "soft %d" % i)
                worst = max(worst, e)
            else:
                assert not r["soft"][j, :148].any() and not r["hard"][j, :148].any()
            assert np.all(r["soft"][j, 148:] == -1.0)          # nothing written past nsoft
    assert worst <= MEASURED, worst


@pytest.mark.parametrize("sps,B,sigmas", [(4, 4096, None), (4, 4096, (0.0, 0.1, 0.5, 2.0)), (2, 1024, None), (1, 2048, None)])
def test_random_batch_vs_oracle_tolerance(pkg, ctx, sps, B, sigmas):
    o = oraclebind.Oracle(sps)
    for tsc in (0, 5):
        kw = {} if sigmas is None else {"sigmas": sigmas}
        x, off, length, meta = synth.normal_batch(sps, B, tsc, seed=1234 + tsc + 10 * sps, **kw)
        gb = GpuBatch(x, off, length, nsoft=148, stride=157)
        ctx[sps].detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, avgpwr=gb.pwr,
                                     hard=gb.hard, energy_thresh=0.0, nsoft=148, soft_stride=157)
        r = gb.results()
        ok, amp, toa, soft = o.normal_batch(x, off, length, tsc, nsoft=148, nthreads=8)
        okb = ok.astype(bool)
        assert_veq((r["flags"] & pkg.F_DETECT) != 0, okb, "detect")
        assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa")
        assert_veq(r["hard"][:, :148], (soft > 0.5).astype(np.uint8), "hard")
        assert not r["soft"][~okb][:, :148].any()
        e, frac = grade(r["soft"][okb][:, :148], soft[okb], "soft")
        assert e <= MEASURED * (4 if sigmas else 1), e         # (noisy bursts: a larger Z, the error scales with it)
        # the rearranged arithmetic did run (or this test grades the exact code against itself)
        assert frac > 0.02, frac
        # a burst whose max|x| * |1/amp| is above 8 must have been handed to the exact code
        for b in np.flatnonzero(okb):
            xs = x[off[b]:off[b] + length[b]]
            a = complex(amp[b]); inv = 1.0 / a
            Z = max(np.abs(xs.real).max(), np.abs(xs.imag).max()) * (abs(inv.real) + abs(inv.imag))
            if Z > 8.001:
                assert_veq(r["soft"][b, :148], soft[b], "burst %d (Z = %.1f) is the exact code's" % (b, Z))


def test_nsoft_above_148_is_exact(pkg, ctx):
    sps, tsc, B = 4, 1, 512
    x, off, length, _ = synth.normal_batch(sps, B, tsc, seed=31)
    gb = GpuBatch(x, off, length, nsoft=156, stride=157)
    ctx[sps].detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, hard=gb.hard, energy_thresh=0.0,
                                 nsoft=156, soft_stride=157)
    r = gb.results()
    ok, amp, toa, soft = oraclebind.Oracle(sps).normal_batch(x, off, length, tsc, nsoft=156, nthreads=8)
    assert_veq(r["soft"][:, :156], soft, "soft")


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def test_hostile_bursts_equal_the_exact_mode(pkg):
    """trxsig_demodulate_batch with the caller's amplitude / TOA: every case the fast form must refuse, beside ordinary ones.
    The exact mode (graded against the oracle here as well, where the oracle's loop ends) is the reference for the bit patterns."""
    import torch
    sps, tsc = 4, 2
    dev = torch.device("cuda:0")
    te = pkg.TrxSig(sps, 0); te.use_torch_stream()
    tt = pkg.TrxSig(sps, 0); tt.use_torch_stream(); tt.set_soft_mode(pkg.SOFT_TOLERANCE)
    base_x, off, length, meta = synth.normal_batch(sps, 64, tsc, seed=77, sigmas=(0.0, 0.05))
    x = base_x.copy()
    amp = np.asarray(meta["amp"], np.complex64).copy()       # the channel the bursts were made with: soft symbols near +-1, Z about 1.5
    toa = np.zeros(64, np.float32)
    rng = np.random.default_rng(3)
    toa[:] = (rng.integers(-3 * 512, 6 * 512, 64) / 512.0).astype(np.float32)   # on the grid, |TOA| small
    cases = {}

    def burst(b):
        return slice(int(off[b]), int(off[b] + length[b]))
    cases["ordinary"] = list(range(0, 16))
    amp[16] = 0; cases["amp zero"] = [16]
    amp[17] = np.nan; cases["amp NaN"] = [17]
    amp[18] = complex(np.inf, 0); cases["amp inf"] = [18]
    toa[19] = np.float32(0.3); cases["TOA off the grid"] = [19]
    toa[20] = np.float32(np.nan); cases["TOA NaN"] = [20]
    toa[21] = np.float32(5000.0); cases["TOA out of range"] = [21]
    # (a TOA with a real fraction: with |frac| <= 0.01 delayVector does not filter and three samples in four are never read)
    x[burst(22)][100] = complex(np.nan, 0); toa[22] = np.float32(0.5); cases["a NaN sample"] = [22]
    x[burst(23)][200] = complex(0, np.inf); toa[23] = np.float32(1.25); cases["an infinite sample"] = [23]
    x[burst(24)] = 0; cases["all-zero samples (every soft symbol on the slicer's 0.5)"] = [24]
    x[burst(25)] *= np.float32(1e-20); amp[25] *= np.float32(1e-20); cases["tiny samples, tiny amplitude"] = [25]
    x[burst(26)] *= np.float32(1e20); amp[26] *= np.float32(1e20); cases["huge samples, huge amplitude"] = [26]
    amp[27] *= np.float32(0.01); cases["Z far above 8"] = [27]
    toa[28] = np.float32(-2.0); cases["integer delay (no filter)"] = [28]
    toa[29] = np.float32(1.00390625); cases["fraction 0.996"] = [29]
    toa[30] = np.float32(14.5); cases["samples fall off the front of the staging area"] = [30]    # (an access burst's kind of delay)
    toa[35] = np.float32(200.25); cases["samples fall off the front of the staging area"].append(35)
    toa[31] = np.float32(-3.5); cases["the first soft symbol reads before the burst"] = [31]
    toa[32] = np.float32(-0.001953125); cases["fraction 1/512: below the filter threshold"] = [32]
    x[burst(33)] *= np.float32(1e-30); amp[33] *= np.float32(1e-30); cases["1/amp beyond 1e15"] = [33]
    x[burst(34)][::2] = 0; cases["every other sample zero"] = [34]
    cases["ordinary, second half"] = list(range(36, 64))

    def run(t):
        gb = GpuBatch(x, off, length, nsoft=148, stride=148)
        a = torch.from_numpy(amp.view(np.float32).reshape(-1, 2).copy()).to(dev)
        to = torch.from_numpy(toa.copy()).to(dev)
        t.demodulate(gb.x, gb.off, gb.len, a, to, gb.soft, hard=gb.hard, nsoft=148, soft_stride=148)
        r = gb.results()
        return r["soft"], r["hard"]

    se, he = run(te)
    st, ht = run(tt)
    assert np.array_equal(he, ht), "hard bits"
    must_be_exact = [k for k in cases if k not in ("ordinary", "ordinary, second half", "integer delay (no filter)", "fraction 0.996",
                                                   "the first soft symbol reads before the burst", "fraction 1/512: below the filter threshold",
                                                   "every other sample zero", "samples fall off the front of the staging area")]
    for k in must_be_exact:
        for b in cases[k]:
            assert np.array_equal(_bits(se[b]), _bits(st[b])), k
    for k in cases:
        for b in cases[k]:
            if k in must_be_exact:
                continue
            d = np.abs(se[b].astype(np.float64) - st[b].astype(np.float64))
            assert not np.isnan(d).any() and d.max() <= BOUND, (k, d.max())
    # (and the ordinary bursts did take the fast form)
    diff = sum(int((_bits(se[b]) != _bits(st[b])).sum()) for b in cases["ordinary"])
    assert diff > 0
    # the exact mode against the oracle on the cases the oracle's loops finish on
    o = oraclebind.Oracle(sps)
    for k in ("ordinary", "integer delay (no filter)", "fraction 0.996", "the first soft symbol reads before the burst",
              "samples fall off the front of the staging area", "Z far above 8", "every other sample zero"):
        for b in cases[k]:
            ref = o.demodulate(x[burst(b)], amp[b], toa[b])[:148]
            assert_veq(se[b][:len(ref)], ref, k)


def test_rach_tolerance(pkg):
    sps, B = 4, 1024
    t = pkg.TrxSig(sps, 0); t.use_torch_stream(); t.set_soft_mode(pkg.SOFT_TOLERANCE)
    x, off, length, meta = synth.rach_batch(sps, B, seed=55)
    gb = GpuBatch(x, off, length, nsoft=148, stride=148)
    t.detect_demod_rach(gb.x, gb.off, gb.len, gb.flags, gb.amp, gb.toa, gb.soft, hard=gb.hard, detect_thresh=5.0, energy_thresh=-1.0,
                        nsoft=148, soft_stride=148)
    r = gb.results()
    ok, amp, toa, soft = oraclebind.Oracle(sps).rach_batch(x, off, length, nthreads=8)
    okb = ok.astype(bool)
    assert_veq((r["flags"] & pkg.F_DETECT) != 0, okb, "detect")
    assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa")
    assert_veq(r["hard"][:, :148], (soft[:, :148] > 0.5).astype(np.uint8), "hard")
    e, frac = grade(r["soft"][okb][:, :148], soft[okb][:, :148], "soft")
    assert e <= 4 * MEASURED and frac > 0.02, (e, frac)


def test_full_batch_sample_tolerance(pkg):
    """BASELINE config 2 at its full size in tolerance mode: a 1,024-burst sample against the oracle, the whole batch against
    the exact mode (hard bits, flags, amp, TOA identical; soft bits within the bound), deterministic to the bit."""
    import torch
    from openbts_ttsou_amd import synth as gsynth
    dev = torch.device("cuda:0")
    sps, tsc, B, NS = 4, 2, 65536, 148
    x, off, length, meta = gsynth.normal_batch_torch(sps, B, tsc, seed=77, device=dev)
    xf = torch.view_as_real(x).contiguous()
    t = pkg.TrxSig(sps, 0); t.use_torch_stream(); t.reserve(B)

    def run():
        r = dict(flags=torch.zeros(B, dtype=torch.uint8, device=dev), amp=torch.zeros(B, 2, device=dev), toa=torch.zeros(B, device=dev),
                 soft=torch.full((B, NS), -1.0, device=dev), hard=torch.zeros((B, NS), dtype=torch.uint8, device=dev))
        t.detect_demod_normal(xf, off, length, tsc, r["flags"], r["amp"], r["toa"], r["soft"], hard=r["hard"], detect_thresh=3.0,
                              energy_thresh=0.0, nsoft=NS, soft_stride=NS)
        torch.cuda.synchronize()
        return r
    exact = run()
    t.set_soft_mode(pkg.SOFT_TOLERANCE)
    tol = run()
    tol2 = run()
    for k in ("flags", "amp", "toa", "hard"):
        assert torch.equal(exact[k], tol[k]), k
    assert torch.equal(tol["soft"].view(torch.int32), tol2["soft"].view(torch.int32))
    d = (exact["soft"].double() - tol["soft"].double()).abs()
    assert float(d.max().item()) <= MEASURED, float(d.max().item())
    assert float((d > 0).float().mean().item()) > 0.02
    rng = np.random.default_rng(9)
    pick = torch.from_numpy(np.sort(rng.choice(B, 1024, replace=False))).to(dev)
    offs = off[pick].cpu().numpy().astype(np.int64); lens = length[pick].cpu().numpy().astype(np.int64)
    xs = np.concatenate([xf[o:o + n].cpu().numpy().view(np.complex64).ravel() for o, n in zip(offs, lens)])
    so = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    ok, amp, toa, soft = oraclebind.Oracle(sps).normal_batch(xs, so, lens.astype(np.int32), tsc, nsoft=NS, nthreads=8)
    okb = ok.astype(bool)
    assert_veq(((tol["flags"][pick] & pkg.F_DETECT) != 0).cpu().numpy(), okb, "detect")
    assert_veq(tol["amp"][pick].cpu().numpy().view(np.complex64).ravel(), amp, "amp")
    assert_veq(tol["toa"][pick].cpu().numpy(), toa, "toa")
    assert_veq(tol["hard"][pick].cpu().numpy(), (soft > 0.5).astype(np.uint8), "hard")
    e, _ = grade(tol["soft"][pick].cpu().numpy()[okb], soft[okb], "soft")
    assert e <= MEASURED, e


def test_group_on_the_fused_front_end_tolerance(pkg):
    """The Transceiver group on the fused receive front end (config 4's default call: schedule with access-burst slots, per-ARFCN
    threshold state, k_demod_rx) in tolerance mode against itself in exact mode, three pushes: valid / RSSI / timing / thresholds
    identical, hard bits identical, soft bits within the bound."""
    import torch
    from openbts_ttsou_amd import synth as gsynth
    from openbts_ttsou_amd.frontend import RxFrontEnd
    sps, S, K, tsc, pushes = 4, 16, 25, 2, 3
    dev = torch.device("cuda:0")
    nb = (K * pushes * 585 // 156 + 4 + 3) // 4 * 4
    x, off, length, meta = gsynth.normal_batch_torch(sps, S * nb, tsc, seed=91, device=dev, sigmas=(0.02, 0.05))
    hi = x.reshape(-1)[: S * (x.numel() // S)].reshape(S, -1)
    t = torch.arange(K * pushes * 864, device=dev, dtype=torch.float64) * (65.0 * sps / 96.0)
    i0 = t.floor().long().clamp(max=hi.shape[1] - 2); fr = (t - i0).to(torch.float32)
    lo = hi[:, i0] * (1 - fr) + hi[:, i0 + 1] * fr
    lo = lo * (8000.0 / lo.abs().amax(dim=1, keepdim=True))
    iq = torch.stack([lo.imag, lo.real], dim=2).round().clamp(-32768, 32767).to(torch.int16).contiguous()
    lpf = gsynth.design_lpf(961, 65 * sps)
    outs = []
    for mode in (pkg.SOFT_EXACT, pkg.SOFT_TOLERANCE):
        ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream(); ctx.set_soft_mode(mode)
        g = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD, start=(0, 0))
        fe = RxFrontEnd(ctx, S, lpf, max_chunks=K)
        for a in range(S):
            g.control(a, "CMD SETTSC %d" % tsc)
            for tn in range(8):
                g.control(a, "CMD SETSLOT %d %d" % (tn, 5 if (tn == 0 and a % 4 == 0) else 1))
        got, slots = [], 0
        for p in range(pushes):
            ns, _ = g.pull_rxfe(fe, iq[:, p * K * 864:(p + 1) * K * 864], (slots // 8))
            got.append(g.collect()); slots += ns
        outs.append(got)
        fe.close(); g.close(); ctx.close()
    n_valid = 0
    worst = 0.0
    for a, b in zip(*outs):
        for key in ("valid", "rssi", "timing", "threshold"):
            assert np.array_equal(a[key], b[key], equal_nan=True), key
        v = a["valid"]
        n_valid += int(v.sum())
        assert np.array_equal(a["soft"][v] > 0.5, b["soft"][v] > 0.5)
        d = np.abs(a["soft"][v].astype(np.float64) - b["soft"][v])
        assert d.max() <= BOUND
        worst = max(worst, float(d.max()))
        assert np.array_equal(a["soft"][~v], b["soft"][~v])
    assert n_valid > 500 and 0 < worst <= 4 * MEASURED, (n_valid, worst)
