"""Side measurement, TX chain (rows a8, a22): modulateBurst for 65,536 bursts (sps 4, guard 8) and the 96:260 / 260:96
polyphase resamplers on a 4 M-sample stream.   python tools/tx_bench.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
dev = torch.device('cuda:0')
B, sps = 65536, 4
t = pkg.TrxSig(sps, 0); t.use_torch_stream()
g = torch.Generator(device=dev); g.manual_seed(1)
bits = torch.randint(0, 2, (B, 148), dtype=torch.uint8, device=dev, generator=g)
guard = torch.full((B,), 8, dtype=torch.int32, device=dev)
n = sps * 156
off = (torch.arange(B, dtype=torch.int32, device=dev) * n).contiguous()
x = torch.zeros(B * n, 2, device=dev)


def timeit(f, K=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


dt = timeit(lambda: t.modulate(bits, guard, x, off))
out = {'modulate': {'us_per_64k_bursts': round(dt * 1e6, 1), 'Mbursts_per_s': round(B / dt / 1e6, 1),
                    'write_TBps': round(B * n * 8 / dt / 1e12, 2)}}
print(json.dumps(out))
