"""Side measurement, TX chain through the back end object (trxsig_txbe): S ARFCN streams x nb bursts per push + one pop,
fused (one kernel per pop that modulates from the queued bits) against unfused (k_modulate into the complex float32 send
buffer at push, k_resample at pop).   python tools/txbe_bench.py [S] [bursts per push]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
from openbts_ttsou_amd.frontend import TxBackEnd
from openbts_ttsou_amd import synth
dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 480
sps = 4
lpf = synth.design_lpf(651, 96)
g = torch.Generator(device=dev); g.manual_seed(1)
bits = torch.randint(0, 2, (S, nb, 148), dtype=torch.uint8, device=dev, generator=g)
gain = torch.rand(S, nb, device=dev, generator=g) * 0.9 + 0.1
guard = np.array([8 + (k % 4 == 0) for k in range(nb)], np.int32)
out = {}
ref = None
for fused in (True, False):
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    be = TxBackEnd(ctx, S, lpf, max_bursts=nb, fused=fused)

    def step():
        be.push_bursts(bits, guard, gain)
        return be.pop_samples()
    for _ in range(10): iq = step()
    torch.cuda.synchronize()
    K = 100
    t0 = time.perf_counter()
    for _ in range(K): iq = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    ctx.profile_enable(True)
    for _ in range(20): step()
    prof = ctx.profile_collect()
    ctx.profile_enable(False)
    out['fused' if fused else 'unfused'] = {'us_per_step': round(dt * 1e6, 1), 'Mbursts_per_s': round(S * nb / dt / 1e6, 1),
                                            'int16_out_GBps': round(iq.numel() * 2 / dt / 1e9, 1),
                                            'kernels_us': {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.items()}}
    h = iq.clone()
    if ref is None: ref = h
    else: out['same_int16_stream_last_step'] = bool(torch.equal(ref, h))
print(json.dumps(out))
