"""Experiment: config-5 steps (trxsig_equalize_normal_batch_fmt on 65,536 fp16 bursts) alternated over S independent
contexts / streams: do the latency-bound equaliser kernels of different batches fill each other's idle issue slots?
   python tools/config5_streams.py"""
import sys, time, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
import _pkg
pkg = _pkg.load()
import bench_config5
dev = torch.device('cuda:0')
args = types.SimpleNamespace(bursts=65536)


def run(S, K=300, W=50):
    ws = []
    for i in range(S):
        st = torch.cuda.Stream()
        c = pkg.TrxSig(1, 0); c.set_stream(st.cuda_stream)
        w = bench_config5.Config5(args)
        with torch.cuda.stream(st):
            w.setup(pkg, c, dev, i, args)
        ws.append(w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04: ws[0].step()
    for i in range(W): ws[i % S].step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): ws[i % S].step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print('%d streams: %.1f us/step  %.1f Mbursts/s' % (S, dt * 1e6, 65536 / dt / 1e6), flush=True)


for S in (1, 2, 3, 4):
    run(S)
