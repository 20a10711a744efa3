"""Where does k_tsc_peak2's time go?  Needs a library whose trxsig_normal.hip was built with -DTRX_PEAK_PROBE
(clock64() stamps of every 16th burst's lanes come back through avgpwr):
    make -C openbts-ttsou_amd/csrc probe && TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_probe.so python tools/peak_probe.py"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
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
toa = torch.zeros(B, device=dev); ap = torch.zeros(B, device=dev)
for _ in range(300):
    c.detect_demod_normal(xf, off, length, 2, flags, amp, toa, None, avgpwr=ap, nsoft=0, soft_stride=0)
torch.cuda.synchronize()
v = ap.cpu().numpy().reshape(-1, 16).astype(np.float64)
names = ['start', 'loads landed, table staged', 'barrier passed', 'bisection + final point', 'tail + stores issued']
prev = 0
for k in range(1, len(names)):
    m = v[:, k].mean()
    print('%-28s %9.0f cycles  (+%7.0f)   min %8.0f max %8.0f' % (names[k], m, m - prev, v[:, k].min(), v[:, k].max()))
    prev = m
