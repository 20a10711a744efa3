"""GPU parity, access (RACH) bursts: detectRACHBurst + demodulateBurst through the C-ABI against the
golden vectors of the real reference and the CPU oracle on random batches.  Value-exact."""
import numpy as np
import pytest

import _pkg
import oraclebind
import synth
from util import GpuBatch, assert_veq

pytestmark = pytest.mark.gpu


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
    return c


@pytest.mark.parametrize("name", ["rach_sps4.npz", "rach_sps1.npz"])
def test_golden_rach(pkg, ctx, golden, name):
    g = golden(name)
    sps = int(g["sps"]); t = ctx[sps]
    gb = GpuBatch(g["x"], g["off"], g["len"], nsoft=148, stride=148)
    t.detect_demod_rach(gb.x, gb.off, gb.len, gb.flags, gb.amp, gb.toa, gb.soft, avgpwr=gb.pwr, hard=gb.hard,
                        detect_thresh=5.0, energy_thresh=-1.0)
    r = gb.results()
    det = (r["flags"] & pkg.F_DETECT) != 0
    assert_veq(det, g["ok"].astype(bool), "detect flags")
    assert_veq(r["amp"], g["amp"], "amp"); assert_veq(r["toa"], g["toa"], "toa")
    for i in range(len(det)):
        if det[i]:
            assert_veq(r["soft"][i], g["soft"][i, :148], "soft %d" % i)
            assert_veq(r["hard"][i], (g["soft"][i, :148] > 0.5).astype(np.uint8))
        else:
            assert not r["soft"][i].any()


@pytest.mark.parametrize("sps,B", [(4, 1024), (2, 256), (1, 512)])
def test_random_rach_vs_oracle(pkg, ctx, sps, B):
    o = oraclebind.Oracle(sps)
    x, off, length, meta = synth.rach_batch(sps, B, seed=4321 + sps, sigmas=(0.0, 0.1, 0.3, 3.0))
    gb = GpuBatch(x, off, length, nsoft=148, stride=148)
    ctx[sps].detect_demod_rach(gb.x, gb.off, gb.len, gb.flags, gb.amp, gb.toa, gb.soft, hard=gb.hard,
                               energy_thresh=-1.0)   # the oracle batch has no energy gate (late bursts start silent)
    r = gb.results()
    ok, amp, toa, soft = o.rach_batch(x, off, length, nthreads=8)
    assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "detect")
    assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa")
    assert_veq(r["soft"], soft, "soft")
    # physics: clean access bursts are found at their true delay and demodulate to the sent payload
    clean = np.flatnonzero(ok.astype(bool) & (meta["sigma"] <= 0.1))
    assert len(clean) > B // 4
    assert np.abs(r["toa"][clean] - meta["delay"][clean]).max() < 0.75
    assert np.array_equal(r["hard"][clean][:, 8:85], meta["bits"][clean][:, 8:85])


def test_host_wrappers(pkg, ctx, golden):
    """The PCIe-inclusive host-buffer entry points give the same answers."""
    g = golden("rach_sps4.npz")
    r = ctx[4].detect_demod_host(g["x"], g["off"], g["len"], tsc=None, energy_thresh=-1.0)
    assert_veq((r["flags"] & pkg.F_DETECT) != 0, g["ok"].astype(bool))
    assert_veq(r["amp"], g["amp"]); assert_veq(r["toa"], g["toa"])
    g = golden("normal_sps4.npz")
    sel = np.flatnonzero(g["tsc"] == 3)
    x = np.concatenate([g["x"][o:o + n] for o, n in zip(g["off"][sel], g["len"][sel])])
    ln = g["len"][sel]; off = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.int32)
    r = ctx[4].detect_demod_host(x, off, ln, tsc=3, energy_thresh=-1.0)
    assert_veq((r["flags"] & pkg.F_DETECT) != 0, g["ok"][sel].astype(bool))
    assert_veq(r["amp"], g["amp"][sel]); assert_veq(r["toa"], g["toa"][sel])
    det = (r["flags"] & pkg.F_DETECT) != 0
    assert_veq(r["soft"][det], g["soft"][sel][det][:, :148])


@pytest.mark.parametrize("variant", ["0", "1", "2"])
def test_rach_variants_agree_with_oracle(pkg, variant, monkeypatch):
    """The RACH routes (0: exact at every lag, 1: approximate-then-exact in one kernel, 2: the same with the bisection
    in its own two-lanes-per-burst kernel and a hand-over list) against the oracle, including noise-only, silent,
    clipped and late bursts where the approximate pass has to hand over."""
    monkeypatch.setenv("TRXSIG_RACH_VARIANT", variant)
    sps, B = 4, 768
    t = pkg.TrxSig(sps, 0, tuning=True); t.use_torch_stream()      # route 0 (exact at every lag) is only in the tuning build
    o = oraclebind.Oracle(sps)
    x, off, length, meta = synth.rach_batch(sps, B, seed=977, sigmas=(0.0, 0.05, 0.3, 1.0, 5.0), max_delay_sym=100)
    rng = np.random.default_rng(3)
    for i in range(0, B, 16):                       # noise only
        x[off[i]:off[i] + length[i]] = (rng.standard_normal(length[i]) + 1j * rng.standard_normal(length[i])) * 40
    for i in range(5, B, 64):                       # silence
        x[off[i]:off[i] + length[i]] = 0
    for i in range(7, B, 64):                       # constant (flat correlation)
        x[off[i]:off[i] + length[i]] = 100 + 50j
    gb = GpuBatch(x, off, length)
    t.detect_demod_rach(gb.x, gb.off, gb.len, gb.flags, gb.amp, gb.toa, gb.soft, energy_thresh=-1.0)
    r = gb.results()
    ok, amp, toa, soft = o.rach_batch(x, off, length, nthreads=8)
    assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "detect")
    assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa"); assert_veq(r["soft"], soft, "soft")
    # a threshold placed exactly on bursts' own peak-to-valley ratios forces the exact-valley hand-over
    ptm = np.array([o.detect_rach(x[off[i]:off[i] + length[i]])["peak_to_mean"] for i in range(64)], np.float32)
    for i in np.flatnonzero(ptm > 0)[:12]:
        for thr in (ptm[i], np.nextafter(ptm[i], np.float32(0)), np.nextafter(ptm[i], np.float32(1e9))):
            gb1 = GpuBatch(x[off[i]:off[i] + length[i]], [0], [length[i]])
            t.detect_demod_rach(gb1.x, gb1.off, gb1.len, gb1.flags, gb1.amp, gb1.toa, gb1.soft, detect_thresh=float(thr),
                                energy_thresh=-1.0)
            want = o.detect_rach(x[off[i]:off[i] + length[i]], thresh=float(thr))["ok"]
            assert bool(gb1.results()["flags"][0] & pkg.F_DETECT) == want, (i, thr)


def _wide_ptm_batch(sps, B, seed):
    """Access bursts whose peak-to-valley ratio spans 1 .. >1000: the usual noise levels, plus bursts whose samples
    after the synch sequence are attenuated (clean valley -> large ratio), noise-only, silent and constant bursts."""
    x, off, length, meta = synth.rach_batch(sps, B, seed=seed, sigmas=(0.0, 0.02, 0.05, 0.1, 0.3, 1.0, 3.0), max_delay_sym=40)
    rng = np.random.default_rng(seed + 1)
    for i in range(0, B, 3):                         # quiet tail: everything from 10 symbols after the synch sequence on
        d = int(meta["delay"][i])
        k = d + (49 + 10) * sps
        x[off[i] + k: off[i] + length[i]] *= np.float32(10.0 ** -rng.uniform(0.0, 3.5))
    for i in range(1, B, 61):                        # noise only
        x[off[i]:off[i] + length[i]] = (rng.standard_normal(length[i]) + 1j * rng.standard_normal(length[i])) * 40
    for i in range(2, B, 127):
        x[off[i]:off[i] + length[i]] = 0
    for i in range(4, B, 127):                       # flat correlation: small maximum against a large energy
        x[off[i]:off[i] + length[i]] = 100 + 50j
    for i in range(6, B, 127):                       # a tone: ditto
        n = np.arange(length[i]); x[off[i]:off[i] + length[i]] = (700 * np.exp(2j * np.pi * 0.013 * n)).astype(np.complex64)
    return x, off, length, meta


@pytest.mark.parametrize("variant", ["1", "2"])
def test_detect_flag_over_thresholds(pkg, variant, monkeypatch):
    """The approximate-then-exact routes decide from an approximate valley unless the threshold falls inside its error
    bar (rach_decide, trxsig_rach.hip).  The bar grows with (peak/valley)^2, so sweep the threshold from 1 to 100 over
    4096 bursts whose ratios cover that whole range: the flag must be the reference's for every burst and threshold
    (amplitude and TOA do not depend on the threshold and must stay value-exact too)."""
    monkeypatch.setenv("TRXSIG_RACH_VARIANT", variant)
    sps, B = 4, 4096
    t = pkg.TrxSig(sps, 0); t.use_torch_stream()
    o = oraclebind.Oracle(sps)
    x, off, length, meta = _wide_ptm_batch(sps, B, seed=31337)
    gb = GpuBatch(x, off, length)
    seen = set()
    for thr in (1.0, 5.0, 13.0, 30.0, 100.0):
        t.detect_demod_rach(gb.x, gb.off, gb.len, gb.flags, gb.amp, gb.toa, gb.soft, detect_thresh=thr, energy_thresh=-1.0)
        r = gb.results()
        ok, amp, toa, soft = o.rach_batch(x, off, length, thresh=thr, nthreads=16)
        assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "detect flags at threshold %g" % thr)
        assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa"); assert_veq(r["soft"], soft, "soft")
        seen.add(int(ok.sum()))
    assert len(seen) >= 4, "the batch does not spread over the thresholds: %r" % (seen,)
    # thresholds placed ON bursts' own ratios (and one ulp either side), 96 bursts across the range
    ptm = np.array([o.detect_rach(x[off[i]:off[i] + length[i]])["peak_to_mean"] for i in range(0, B, 8)], np.float32)
    order = np.argsort(ptm)
    pick = [8 * int(j) for j in order[np.linspace(0, len(order) - 1, 96).astype(int)] if ptm[j] > 0]
    for i in pick:
        p = np.float32(ptm[i // 8])
        for thr in (p, np.nextafter(p, np.float32(0)), np.nextafter(p, np.float32(1e30))):
            gb1 = GpuBatch(x[off[i]:off[i] + length[i]], [0], [length[i]])
            t.detect_demod_rach(gb1.x, gb1.off, gb1.len, gb1.flags, gb1.amp, gb1.toa, gb1.soft, detect_thresh=float(thr),
                                energy_thresh=-1.0)
            want = o.detect_rach(x[off[i]:off[i] + length[i]], thresh=float(thr))["ok"]
            assert bool(gb1.results()["flags"][0] & pkg.F_DETECT) == want, (i, float(thr))
