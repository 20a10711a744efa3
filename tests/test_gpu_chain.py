"""GPU parity of the single-launch normal-burst path (k_normal_chain, TRXSIG_TUNE_NORMAL_PATH 5): detect workgroups
hand over to demodulate workgroups inside one launch.  Beyond the shared checks in test_gpu_normal_fused.py (golden
vectors, oracle batches, hostile and ragged inputs): every lag between the two roles gives the same values, repeated
launches on one context leave the hand-over words clean, batch sizes round the tile/stream edges, and a wait that
runs out is reported (and the three-launch path takes over) instead of hanging or returning silently."""
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


def run(t, x, off, length, tsc, energy_thresh=0.0, detect_thresh=3.0, nsoft=148, stride=148):
    gb = GpuBatch(x, off, length, nsoft=nsoft, stride=stride)
    t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, avgpwr=gb.pwr, hard=gb.hard,
                          detect_thresh=detect_thresh, energy_thresh=energy_thresh, nsoft=nsoft, soft_stride=stride)
    return gb.results()


def same(r, q, what):
    for k in ("flags", "amp", "toa", "pwr", "soft", "hard"):
        assert_veq(r[k], q[k], "%s: %s" % (what, k))


def ctx(pkg, sps, path, **kw):
    t = pkg.TrxSig(sps, 0, tuning=(path != 0))               # the chain lives in libtrxsig_tune.so; path 0 is the product library
    t.use_torch_stream()
    if path != 0 or kw:
        t.set_tuning(normal_path=path, **kw)
    return t


@pytest.mark.parametrize("sps", [1, 2, 4])
def test_every_lag_and_batch_edge(pkg, sps):
    ref = ctx(pkg, sps, 0)
    for B in (1, 15, 16, 17, 127, 128, 129, 640, 641, 2049):
        x, off, length, meta = synth.normal_batch(sps, B, 4, seed=100 + B)
        # a third of the bursts silent: their demodulators must write zeros, not wait for anything else
        for b in range(0, B, 3):
            x[off[b]:off[b] + length[b]] *= np.float32(1e-4)
        want = run(ref, x, off, length, 4, energy_thresh=3.0)
        for lag in (1, 2, 5, 48, 100000):
            t = ctx(pkg, sps, 5, chain_lag=lag)
            same(run(t, x, off, length, 4, energy_thresh=3.0), want, "B %d lag %d" % (B, lag))
            same(run(t, x, off, length, 4, energy_thresh=3.0), want, "B %d lag %d, second launch" % (B, lag))


def test_large_batch_vs_oracle_and_repeated_launches(pkg):
    sps, B, tsc = 4, 40000, 6
    x, off, length, meta = synth.normal_batch(sps, B, tsc, seed=2024)
    t = ctx(pkg, sps, 5)
    r = run(t, x, off, length, tsc)
    ok, amp, toa, soft = oraclebind.Oracle(sps).normal_batch(x, off, length, tsc, nsoft=148, nthreads=16)
    assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "detect")
    assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa")
    assert_veq(r["soft"], soft[:, :148], "soft")
    assert_veq(r["hard"], (soft[:, :148] > 0.5).astype(np.uint8), "hard")
    # same context, different inputs in the same buffers' place: stale granules would show as the old values
    x2, off2, length2, _ = synth.normal_batch(sps, B, tsc, seed=2025)
    want2 = run(ctx(pkg, sps, 0), x2, off2, length2, tsc)
    for _ in range(3):
        same(run(t, x2, off2, length2, tsc), want2, "second input")
    same(run(t, x, off, length, tsc), r, "first input again")


def test_wait_that_runs_out_is_reported_and_falls_back(pkg):
    """chain_spin 0 + lag 1: a demodulator whose detection is not there at its first look gives up.  The library must
    say so at its next entry (not hang, not hand back the incomplete call silently) and then work through the
    three-launch path, with clean hand-over words."""
    sps, B, tsc = 4, 8192, 1
    x, off, length, meta = synth.normal_batch(sps, B, tsc, seed=7)
    want = run(ctx(pkg, sps, 0), x, off, length, tsc)
    t = ctx(pkg, sps, 5, chain_lag=1, chain_spin=0)
    gb = GpuBatch(x, off, length, nsoft=148, stride=148)
    def call():
        t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, avgpwr=gb.pwr, hard=gb.hard,
                              detect_thresh=3.0, energy_thresh=0.0, nsoft=148, soft_stride=148)
    call()
    import torch
    torch.cuda.synchronize()
    r = gb.results()
    if np.array_equal(r["soft"], want["soft"]):
        pytest.skip("every demodulator found its detection at the first look on this box")
    with pytest.raises(pkg.TrxSigError, match="timed out"):
        call()
    call()                                                    # the fallback path
    same(gb.results(), want, "after the fallback")
