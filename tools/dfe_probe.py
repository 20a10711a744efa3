"""Who waits for whom in k_eq_dfe2?  Needs trxsig_eq.hip built with -DTRX_DFE_PROBE (barrier-wait cycles of the producer and the
consumer wave and each wave's lifetime come back in the first four soft values of every 64th burst):
    TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_dfeprobe.so python tools/dfe_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
from openbts_ttsou_amd import synth
dev = torch.device('cuda:0')
B, tsc = 65536, 6
x, off, length, meta = synth.normal_batch_torch(1, B, tsc, seed=5, device=dev, sigmas=(0.02, 0.1), max_delay=1.0)
xf = torch.view_as_real(x).contiguous()
t = pkg.TrxSig(1, 0); t.use_torch_stream(); t.reserve(B)
d = dict(flags=torch.zeros(B, dtype=torch.uint8, device=dev), amp=torch.zeros(B, 2, device=dev), toa=torch.zeros(B, device=dev),
         w=torch.zeros(B, 7, 2, device=dev), b=torch.zeros(B, 5, 2, device=dev), soft=torch.zeros(B, 157, device=dev))
for _ in range(50):
    t.equalize_normal(xf, off, length, tsc, d['flags'], d['amp'], d['toa'], d['soft'], w=d['w'], b=d['b'], energy_thresh=10.0,
                      variant52m=True, max_toa=4, nsoft=156, soft_stride=157)
torch.cuda.synchronize()
v = d['soft'].cpu().numpy()[::64, :4].astype(np.float64)
for name, k in (('producer: cycles waiting at barriers', 0), ('producer: lifetime', 1), ('consumer: cycles waiting at barriers', 2), ('consumer: lifetime', 3)):
    print('%-40s mean %8.0f  p10 %8.0f  p90 %8.0f' % (name, v[:, k].mean(), np.percentile(v[:, k], 10), np.percentile(v[:, k], 90)))
