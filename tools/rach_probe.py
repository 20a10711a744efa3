"""Where does k_rach_fast's time go?  Needs the probe build (make -C openbts-ttsou_amd/csrc probe):
    TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_probe.so python tools/rach_probe.py"""
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
x, off, length, meta = synth.rach_batch_torch(4, B, seed=1, device=dev)
xf = torch.view_as_real(x).contiguous()
c = pkg.TrxSig(4, 0); c.use_torch_stream(); c.reserve(B)
flags = torch.zeros(B, dtype=torch.uint8, device=dev); amp = torch.zeros(B, 2, device=dev)
toa = torch.zeros(B, device=dev); ap = torch.zeros(B, device=dev)
for _ in range(20):
    c.detect_demod_rach(xf, off, length, flags, amp, toa, None, avgpwr=ap, detect_thresh=5.0, energy_thresh=-1.0, nsoft=0, soft_stride=0)
torch.cuda.synchronize()
v = ap.cpu().numpy().reshape(-1, 8).astype(np.float64)
names = ['start', 'burst staged, energy', 'pulse filter', 'approximate correlation + argmax', 'exact contenders + neighbourhood',
         'bisection (split route: record written)', 'tail']
prev = 0
for k in range(1, 6 if (v[:, 6] <= 0).all() else 7):     # the split route stops after stamp 5
    m = v[:, k].mean()
    print('%-34s %9.0f cycles  (+%7.0f)   p10 %8.0f p90 %8.0f' % (names[k], m, m - prev, np.percentile(v[:, k], 10), np.percentile(v[:, k], 90)))
    prev = m
