"""Experiment: does processing the batch in chunks on ONE stream (so that a chunk's correlation window is still in
the memory-side cache when its demodulation re-reads it) pay?  Also: rotating distinct inputs (no cross-step reuse)
and the demodulation alone.   python tools/chunk_locality.py"""
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
c = pkg.TrxSig(4, 0); c.use_torch_stream(); c.reserve(B)
flags = torch.zeros(B, dtype=torch.uint8, device=dev); amp = torch.zeros(B, 2, device=dev)
toa = torch.zeros(B, device=dev); soft = torch.zeros(B, 148, device=dev)


def timeit(step, K=1000, W=200, label=''):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04: step(0)
    for i in range(W): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print('%-34s %.1f us/step  %.1f Mbursts/s' % (label, dt * 1e6, B / dt / 1e6), flush=True)


def chunked(n):
    h = B // n
    sl = [slice(i * h, (i + 1) * h) for i in range(n)]
    def step(_):
        for s in sl:
            c.detect_demod_normal(xf, off[s], length[s], 2, flags[s], amp[s], toa[s], soft[s], nsoft=148, soft_stride=148)
    return step


for n in (1, 2, 4):
    timeit(chunked(n), label='%d chunk(s), one stream' % n)
xs = [xf] + [xf.clone() for _ in range(3)]
timeit(lambda i: c.detect_demod_normal(xs[i % 4], off, length, 2, flags, amp, toa, soft, nsoft=148, soft_stride=148),
       label='4 rotating inputs (1.3 GB)')
c.detect_demod_normal(xf, off, length, 2, flags, amp, toa, soft, nsoft=148, soft_stride=148)
timeit(lambda i: c.demodulate(xf, off, length, amp, toa, soft, enable=flags, nsoft=148, soft_stride=148), label='demodulate alone, same input')
timeit(lambda i: c.demodulate(xs[i % 4], off, length, amp, toa, soft, enable=flags, nsoft=148, soft_stride=148), label='demodulate alone, rotating inputs')
timeit(lambda i: c.detect_demod_normal(xf, off, length, 2, flags, amp, toa, None, nsoft=0, soft_stride=0), label='detect alone')
