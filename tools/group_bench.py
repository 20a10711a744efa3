"""Throughput of the Transceiver group (trxsig_trxgroup_pull) against the one-burst-per-call object: S ARFCNs, all eight
timeslots combination I except TN 0 of every eighth ARFCN (combination V), clean bursts.  Prints per-call time for calls of
1 frame (8 slots) up to 64 frames.  Usage: python tools/group_bench.py [sps] [S]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
import synth

sps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
leg = pkg.TSCLEG_DEMOD if sps != 1 else pkg.TSCLEG_EQUALIZE
cell = 160 * sps
ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
g = pkg.TrxGroup(ctx, S, tsc_leg=leg)
for a in range(S):
    g.control(a, "CMD SETTSC 2")
    for tn in range(8):
        g.control(a, "CMD SETSLOT %d %d" % (tn, 5 if (tn == 0 and a % 8 == 0) else 1))
frames = 64
n_slots = 8 * frames
pool = 4096
xs, offs, lens, _ = synth.normal_batch(sps, pool, 2, seed=1, sigmas=(0.0, 0.1), max_delay=1.0)
x = np.zeros((n_slots, S, cell), np.complex64)
rng = np.random.default_rng(0)
for t in range(n_slots):
    n = (156 + (t % 4 == 0)) * sps
    idx = rng.integers(0, pool // 4, S) * 4 + (0 if t % 4 == 0 else 1)
    for a in range(S):
        i = idx[a]
        x[t, a, :min(n, lens[i])] = xs[offs[i]:offs[i] + lens[i]][:n]
dx = torch.from_numpy(x.view(np.float32).reshape(-1)).to("cuda:0")
print("sps %d, S %d, leg %s" % (sps, S, "equalize" if leg == 0 else "demod"))
for fr in (1, 2, 8, 64):
    ns = 8 * fr
    for rep in range(3):
        g.pull(dx, S * cell, cell, 100, 0, ns); torch.cuda.synchronize()
    reps = max(4, 256 // fr)
    t0 = time.perf_counter()
    for rep in range(reps):
        res = g.pull(dx, S * cell, cell, 100 + rep * fr, 0, ns)
    t_sub = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    r = g.collect(soft=False)
    print("  %3d frame(s) per call: %6d rows, submit %7.1f us, end to end %7.1f us per call = %6.3f us per burst (%.1f Mbursts/s), %d valid"
          % (fr, res.n_rows, t_sub / reps * 1e6, t_all / reps * 1e6, t_all / reps / max(res.n_rows, 1) * 1e6,
             res.n_rows * reps / t_all / 1e6, int(r["valid"].sum())))
    ctx.profile_enable(True)
    g.pull(dx, S * cell, cell, 100, 0, ns)
    print("     kernels (us):", {k: round(v[0] * 1e3, 1) for k, v in ctx.profile_collect().items()})
    ctx.profile_enable(False)
# the one-burst-per-call object on the same bursts
h = pkg.TrxHost(sps, 0, tsc_leg=leg)
h.control("CMD SETTSC 2")
for tn in range(8):
    h.control("CMD SETSLOT %d 1" % tn)
t0 = time.perf_counter(); n = 0
for t in range(64):
    nn = (156 + (t % 4 == 0)) * sps
    r = h.pull_radio_vector(x[t, 1, :nn], t % 8, 100 + t // 8); n += 1
print("  one burst per call (trxsig_trx_pull_radio_vector): %.1f us per burst" % ((time.perf_counter() - t0) / n * 1e6))
