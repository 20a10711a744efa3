"""Experiment: whole steps alternated over S independent HIP streams (own context, own intermediates, no
cross-stream events) -- the latency-bound k_tsc_peak and the VALU-bound k_tsc_corr of one step can share the
card with the HBM-bound k_demod of another.   python tools/two_stream_steps.py"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import _pkg
pkg = _pkg.load()
from openbts_ttsou_amd import synth
dev = torch.device('cuda:0')
B = 65536
x, off, length, meta = synth.normal_batch_torch(4, B, 2, seed=1, device=dev)
xf = torch.view_as_real(x).contiguous()


def run(S, K=1000, W=200, distinct_inputs=False):
    st = [torch.cuda.Stream() for _ in range(S)]
    cs = []
    for s in st:
        c = pkg.TrxSig(4, 0); c.set_stream(s.cuda_stream); c.reserve(B); cs.append(c)
    outs = [(torch.zeros(B, dtype=torch.uint8, device=dev), torch.zeros(B, 2, device=dev), torch.zeros(B, device=dev),
             torch.zeros(B, 148, device=dev)) for _ in range(S)]
    xs = [xf.clone() if (distinct_inputs and i) else xf for i in range(S)]
    def step(i):
        k = i % S
        f, a, t, so = outs[k]
        cs[k].detect_demod_normal(xs[k], off, length, 2, f, a, t, so, nsoft=148, soft_stride=148)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04: step(0)
    for i in range(W): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print('%d streams%s: %.1f us/step  %.1f Mbursts/s' % (S, ' (own inputs)' if distinct_inputs else '', dt * 1e6, B / dt / 1e6), flush=True)
    ref = outs[0][3].clone()
    for o in outs[1:]:
        assert torch.equal(o[3], ref)


for S in (1, 2):
    run(S)
for S in (2, 3, 4):
    run(S, distinct_inputs=True)
