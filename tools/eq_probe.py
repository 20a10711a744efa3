"""Where does k_eq_detect52's (k_eq_detect's) time go?  Needs trxsig_eq.hip built with -DTRX_EQ_PROBE (clock64() stamps of the detected
bursts come back through toa):
    make -C openbts-ttsou_amd/csrc probe_eq EQP=1 && TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_eqprobe1.so python tools/eq_probe.py
(EQP=2: the staging block in detail -- rows then read: offsets here / loads issued / landed + parked / read back / end of kernel)"""
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
v = d['toa'].cpu().numpy().reshape(-1, 8).astype(np.float64)
names = ['start', 'energy', 'correlation', 'peakDetect', 'tail + delayVector + channel pick', 'designDFE + taps']
prev = 0
for k in range(1, 6):
    m = v[:, k].mean()
    print('%-36s %9.0f cycles  (+%7.0f)' % (names[k], m, m - prev))
    prev = m

# absolute stamps (24 bits of clock64): how the launch's waves spread over time
st = v[:, 6]; en = v[:, 7]
t0 = st.min()
st = (st - t0) % (1 << 24); en = (en - t0) % (1 << 24)
print('wave start: min 0, median %.0f, p90 %.0f, max %.0f cycles after the first' % (np.median(st), np.percentile(st, 90), st.max()))
print('wave end:   min %.0f, median %.0f, p90 %.0f, max %.0f' % (en.min(), np.median(en), np.percentile(en, 90), en.max()))
print('wave duration: median %.0f, max %.0f' % (np.median(en - st), (en - st).max()))
# the counters of the 8 XCDs are not synchronised: look at each cluster of start times on its own
order = np.argsort(st); ss = st[order]; ee = en[order]
cuts = np.flatnonzero(np.diff(ss) > 200000)
lo = 0
for c in list(cuts) + [len(ss) - 1]:
    a, e = ss[lo:c + 1], ee[lo:c + 1]
    print('  cluster of %5d waves: starts spread over %6.0f cycles (median %6.0f), last end %6.0f after the first start' % (
        len(a), a.max() - a.min(), np.median(a - a.min()), e.max() - a.min()))
    lo = c + 1
