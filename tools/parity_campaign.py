"""A larger parity run than the test-suite's: GPU (through the C-ABI) against the CPU oracle, value-exact, on many
seeds -- normal bursts at every TSC and sps with noise levels up to "undetectable", access bursts, the 52M equaliser
leg.  Run on the GPU box:  python tools/parity_campaign.py [bursts-per-case]   (about a minute with 16 host cores)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch
import _pkg
import oraclebind
from util import GpuBatch, assert_veq
pkg = _pkg.load()
from openbts_ttsou_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NT = min(os.cpu_count() or 1, 16)
t0 = time.time()
total = 0
for sps in (1, 2, 4):
    t = pkg.TrxSig(sps, 0); t.use_torch_stream()
    o = oraclebind.Oracle(sps)
    for tsc in range(8):
        x, off, length, meta = synth.normal_batch(sps, N, tsc, seed=31337 + 100 * sps + tsc, sigmas=(0.0, 0.05, 0.2, 0.5, 1.0, 3.0),
                                                  max_delay=2.5)
        gb = GpuBatch(x, off, length)
        for ethr in (-1.0, 0.0):
            t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, hard=gb.hard, energy_thresh=ethr)
            r = gb.results()
            ok, amp, toa, soft = o.normal_batch(x, off, length, tsc, nthreads=NT)
            assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "detect sps%d tsc%d" % (sps, tsc))
            assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa"); assert_veq(r["soft"], soft, "soft")
        total += N
    x, off, length, meta = synth.rach_batch(sps, N // 2, seed=4242 + sps, sigmas=(0.0, 0.1, 0.3, 1.0, 4.0), max_delay_sym=90)
    gb = GpuBatch(x, off, length)
    t.detect_demod_rach(gb.x, gb.off, gb.len, gb.flags, gb.amp, gb.toa, gb.soft, energy_thresh=-1.0)
    r = gb.results()
    ok, amp, toa, soft = o.rach_batch(x, off, length, nthreads=NT)
    assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "rach detect sps%d" % sps)
    assert_veq(r["amp"], amp, "rach amp"); assert_veq(r["toa"], toa, "rach toa"); assert_veq(r["soft"], soft, "rach soft")
    total += N // 2
    print("sps %d: 8 TSC x %d normal bursts (x2 energy gates) + %d access bursts identical  [%.0f s]" % (sps, N, N // 2, time.time() - t0),
          flush=True)
print("parity campaign: %d bursts, every output value-exact" % total)
