"""GPU parity of the single-kernel normal-burst path (k_normal_fused; TRXSIG_TUNE_NORMAL_PATH 1 and 2):
the same checks as test_gpu_normal.py -- golden vectors captured from the real reference, seeded random
batches against the CPU oracle, ragged / invalid bursts -- plus hostile inputs for the speculative
bisection (all-zero and constant windows, which make the reference leave its early/late loop on equal
powers; noise only; peaks at the window edges).  Everything value-exact (IEEE ==)."""
import numpy as np
import pytest

import _pkg
import oraclebind
import synth
from util import GpuBatch, assert_veq

pytestmark = pytest.mark.gpu

PATHS = [1, 2, 3, 4, 5]    # 5: k_normal_chain (one launch, in-launch hand-over); 1: a wave per burst, 2: two bursts per wave, 3: four bursts per wave (k_normal_quad), 4: k_normal_quad detection + k_demod


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _pkg.load()


@pytest.fixture(scope="module")
def ctxs(pkg):
    c = {}
    for path in [0] + PATHS:
        for s in (1, 2, 4):
            # the alternates live in libtrxsig_tune.so; path 0 -- what they are compared with -- is the PRODUCT library
            t = pkg.TrxSig(s, 0, tuning=(path != 0))
            t.use_torch_stream()
            if path != 0:
                t.set_tuning(normal_path=path)
            c[path, s] = t
    return c


def run(t, x, off, length, tsc, energy_thresh=0.0, detect_thresh=3.0, nsoft=148, stride=160):
    gb = GpuBatch(x, off, length, nsoft=nsoft, stride=stride)
    t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, avgpwr=gb.pwr, hard=gb.hard,
                          detect_thresh=detect_thresh, energy_thresh=energy_thresh, nsoft=nsoft, soft_stride=stride)
    return gb.results()


def same(r, q, what):
    for k in ("flags", "amp", "toa", "pwr", "soft", "hard"):
        assert_veq(r[k], q[k], "%s: %s" % (what, k))


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name", ["normal_sps4.npz", "normal_sps1.npz"])
def test_golden_normal(pkg, ctxs, golden, name, path):
    g = golden(name)
    sps = int(g["sps"]); t = ctxs[path, sps]
    for tsc in range(8):
        sel = np.flatnonzero(g["tsc"] == tsc)
        r = run(t, g["x"], g["off"][sel], g["len"][sel], tsc, energy_thresh=-1.0)
        det = (r["flags"] & pkg.F_DETECT) != 0
        assert_veq(det, g["ok"][sel].astype(bool), "detect flags tsc %d" % tsc)
        assert np.all(r["flags"] & pkg.F_ENERGY)
        assert_veq(r["amp"], g["amp"][sel], "amp"); assert_veq(r["toa"], g["toa"][sel], "toa")
        assert_veq(r["pwr"], g["energy_pwr"][sel], "energyDetect avgPwr")
        for j, i in enumerate(sel):
            if det[j]:
                assert_veq(r["soft"][j, :148], g["soft"][i, :148], "soft %d" % i)
                assert_veq(r["hard"][j, :148], (g["soft"][i, :148] > 0.5).astype(np.uint8), "hard %d" % i)
            else:
                assert not r["soft"][j, :148].any() and not r["hard"][j, :148].any()
            assert np.all(r["soft"][j, 148:] == -1.0)          # nothing written past nsoft
        r2 = run(t, g["x"], g["off"][sel], g["len"][sel], tsc, energy_thresh=float(g["energy_thresh"]))
        e_ok = (r2["flags"] & pkg.F_ENERGY) != 0
        assert_veq(e_ok, g["energy_ok"][sel].astype(bool), "energy flags")
        assert_veq((r2["flags"] & pkg.F_DETECT) != 0, g["ok"][sel].astype(bool) & e_ok)
        assert not r2["amp"][~e_ok].any() and not r2["toa"][~e_ok].any()


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("sps,B", [(4, 4099), (2, 1025), (1, 2050)])
def test_random_batch_vs_oracle(pkg, ctxs, sps, B, path):
    o = oraclebind.Oracle(sps)
    for tsc in (0, 5):
        x, off, length, meta = synth.normal_batch(sps, B, tsc, seed=4321 + tsc + 10 * sps)
        r = run(ctxs[path, sps], x, off, length, tsc, nsoft=148, stride=148)
        ok, amp, toa, soft = o.normal_batch(x, off, length, tsc, nsoft=148, nthreads=8)
        assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "detect")
        assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa")
        assert_veq(r["soft"], soft[:, :148], "soft")
        assert_veq(r["hard"], (soft[:, :148] > 0.5).astype(np.uint8), "hard")
        same(r, run(ctxs[0, sps], x, off, length, tsc, nsoft=148, stride=148), "three-kernel path")


def hostile_batch(sps, tsc, seed):
    """Bursts that stress peakDetect: zeros, constants, noise, peaks moved to the window's edges."""
    rng = np.random.default_rng(seed)
    B = 96
    x, off, length, meta = synth.normal_batch(sps, B, tsc, seed=seed, max_delay=1.5)
    x = x.copy()
    for b in range(B):
        seg = slice(off[b], off[b] + length[b])
        kind = b % 12
        if kind == 0:
            x[seg] = 0                                             # all-zero: argmax -1, "break" at step one
        elif kind == 1:
            x[seg] = 1000.0                                        # constant
        elif kind == 2:
            x[seg] = (rng.standard_normal(length[b]) + 1j * rng.standard_normal(length[b])).astype(np.complex64)
        elif kind == 3:
            x[seg] = np.roll(x[seg], 17 * sps)                     # peak near the end of the window
        elif kind == 4:
            x[seg] = np.roll(x[seg], -9 * sps)                     # peak near / before the start
        elif kind == 5:
            x[seg] = np.roll(x[seg], 18 * sps + 1)
        elif kind == 6:
            y = x[seg].copy(); y[: 60 * sps] = 0; y[90 * sps:] = 0; x[seg] = y     # isolated midamble
        elif kind == 7:
            x[seg] = x[seg] * np.float32(1e-18)                    # powers underflow towards denormals
        elif kind == 8:
            x[seg] = x[seg] * np.float32(1e12)
        elif kind == 9:
            y = np.zeros(length[b], np.complex64); y[66 * sps] = 1.0; x[seg] = y   # a single impulse
    return x, off, length


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("sps", [1, 2, 4])
def test_hostile_windows(pkg, ctxs, sps, path):
    o = oraclebind.Oracle(sps)
    for tsc in (2, 7):
        x, off, length = hostile_batch(sps, tsc, seed=99 + tsc)
        for ethr in (-1.0, 5.0):
            r = run(ctxs[path, sps], x, off, length, tsc, energy_thresh=ethr)
            same(r, run(ctxs[0, sps], x, off, length, tsc, energy_thresh=ethr), "three-kernel path")
        r = run(ctxs[path, sps], x, off, length, tsc, energy_thresh=-1.0, nsoft=148, stride=148)
        ok, amp, toa, soft = o.normal_batch(x, off, length, tsc, nsoft=148, nthreads=4)
        assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "detect")
        assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa")
        assert_veq(r["soft"], soft[:, :148], "soft")


@pytest.mark.parametrize("path", PATHS)
def test_ragged_and_bad_bursts(pkg, ctxs, path):
    sps = 4
    x, off, length, meta = synth.normal_batch(sps, 37, 3, seed=77)
    length2 = length.copy(); off2 = off.copy()
    length2[5] = 91 * sps; length2[6] = 158 * sps; length2[7] = 624 + 1; off2[8] = -4
    off2[9] = off2[9] + 1          # odd offset: the 8-byte load path
    length2[10] = 100 * sps        # short but legal: the tail of the soft bits reads past the burst (zeros)
    length2[11] = 92 * sps
    xx = np.concatenate([x, np.zeros(700, np.complex64)])
    for nsoft, stride in ((148, 160), (100, 100), (0, 1)):
        r = run(ctxs[path, sps], xx, off2, length2, 3, nsoft=nsoft, stride=stride)
        q = run(ctxs[0, sps], xx, off2, length2, 3, nsoft=nsoft, stride=stride)
        if nsoft == 0:
            for k in ("flags", "amp", "toa", "pwr"):
                assert_veq(r[k], q[k], k)
            assert np.all(r["soft"] == -1.0)
        else:
            same(r, q, "nsoft %d" % nsoft)
        assert np.all(r["flags"][[5, 6, 7, 8]] == pkg.F_BADLEN)
    # nsoft > 148 silently takes the three-kernel path; empty batch is a no-op
    same(run(ctxs[path, sps], xx, off2, length2, 3, nsoft=156, stride=157),
         run(ctxs[0, sps], xx, off2, length2, 3, nsoft=156, stride=157), "nsoft 156")
    gb = GpuBatch(xx, off2, length2)
    ctxs[path, sps].detect_demod_normal(gb.x, gb.off[:0], gb.len[:0], 3, gb.flags, gb.amp, gb.toa, gb.soft)


@pytest.mark.parametrize("path", [0, 3, 5])
@pytest.mark.parametrize("sps", [1, 2, 4])
def test_tap_class_specialisation_is_invisible(pkg, sps, path):
    """The correlators' exact-product FMA form (taps with a component of exactly +-1) against the generic
    complex multiply-accumulate: identical outputs, hostile windows included."""
    a = pkg.TrxSig(sps, 0, tuning=True); a.use_torch_stream(); a.set_tuning(normal_path=path, generic_taps=0)
    g = pkg.TrxSig(sps, 0, tuning=True); g.use_torch_stream(); g.set_tuning(normal_path=path, generic_taps=1)
    for tsc in range(8):
        x, off, length, meta = synth.normal_batch(sps, 515, tsc, seed=700 + tsc)
        same(run(a, x, off, length, tsc), run(g, x, off, length, tsc), "tsc %d" % tsc)
        x, off, length = hostile_batch(sps, tsc, seed=800 + tsc)
        same(run(a, x, off, length, tsc, energy_thresh=-1.0), run(g, x, off, length, tsc, energy_thresh=-1.0), "hostile")


@pytest.mark.parametrize("sps", [1, 2, 4])
def test_peak_kernels_are_interchangeable(pkg, sps):
    """Path 0's three peak kernels -- k_tsc_peak2 (two lanes per burst, the default), k_tsc_peak8 (eight lanes,
    speculated bisection) and k_tsc_peak (a lane per burst, the reference's serial loop as written) --
    give identical outputs on random, ragged and hostile batches, tie-breaking bisections included."""
    ctx = []
    for sp in (2, 0, 1):
        c = pkg.TrxSig(sps, 0, tuning=True); c.use_torch_stream(); c.set_tuning(normal_path=0, spec_peak=sp); ctx.append(c)
    for tsc in range(8):
        x, off, length, meta = synth.normal_batch(sps, 1031, tsc, seed=900 + tsc)
        want = run(ctx[0], x, off, length, tsc)
        for c in ctx[1:]:
            same(run(c, x, off, length, tsc), want, "tsc %d" % tsc)
        x, off, length = hostile_batch(sps, tsc, seed=950 + tsc)
        for ethr in (-1.0, 5.0):
            want = run(ctx[0], x, off, length, tsc, energy_thresh=ethr)
            for c in ctx[1:]:
                same(run(c, x, off, length, tsc, energy_thresh=ethr), want, "hostile")
