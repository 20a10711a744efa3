"""Experiment: chunked two-stage pipelining of the normal-burst leg across two HIP streams --
detection (k_tsc_corr + k_tsc_peak) of chunk i+1 runs while chunk i is demodulated (k_demod).
Run on the GPU box:  python tools/overlap_probe.py"""
import sys, time
sys.path.insert(0, '.')
import torch
import _pkg
pkg = _pkg.load()
from openbts_ttsou_amd import synth
dev = torch.device('cuda:0')
B = 65536
x, off, length, meta = synth.normal_batch_torch(4, B, 2, seed=1, device=dev)
xf = torch.view_as_real(x).contiguous()
flags = torch.zeros(B, dtype=torch.uint8, device=dev); amp = torch.zeros(B, 2, device=dev)
toa = torch.zeros(B, device=dev); soft = torch.zeros(B, 148, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
cA = pkg.TrxSig(4, 0); cA.set_stream(sA.cuda_stream); cA.reserve(B)
cB = pkg.TrxSig(4, 0); cB.set_stream(sB.cuda_stream); cB.reserve(B)
en = torch.ones(B, dtype=torch.uint8, device=dev)

def run(nchunk, K=1000, W=200):
    h = B // nchunk
    evs = [torch.cuda.Event() for _ in range(nchunk)]
    done = torch.cuda.Event()
    def step():
        sA.wait_event(done)                      # next step's detection may not overtake this step's demod (same buffers)
        for i in range(nchunk):
            sl = slice(i * h, (i + 1) * h)
            cA.detect_demod_normal(xf, off[sl], length[sl], 2, flags[sl], amp[sl], toa[sl], None, nsoft=0, soft_stride=0)
            evs[i].record(sA)
            sB.wait_event(evs[i])
            cB.demodulate(xf, off[sl], length[sl], amp[sl], toa[sl], soft[sl], enable=flags[sl], nsoft=148, soft_stride=148)
        done.record(sB)
    for _ in range(W): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print('%d chunks: %.1f us/step  %.1f Mbursts/s' % (nchunk, dt * 1e6, B / dt / 1e6), flush=True)

for n in (1, 2, 4, 8, 16):
    run(n)
