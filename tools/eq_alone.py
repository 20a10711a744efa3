"""k_eq_detect launched alone, back to back (trxsig_channel_estimate_batch): its time when no other kernel shares the
instruction caches, against its time inside the three-kernel equaliser step (bench.py --workload config5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import _pkg
pkg = _pkg.load()
from openbts_ttsou_amd import synth
dev = torch.device('cuda:0')
tsc = 6
for B in (16384, 65536):
    x, off, length, meta = synth.normal_batch_torch(1, B, tsc, seed=5, device=dev, sigmas=(0.02, 0.1), max_delay=1.0)
    xf = (torch.view_as_real(x) * 500).contiguous()
    t = pkg.TrxSig(1, 0); t.use_torch_stream(); t.reserve(B)
    fl = torch.zeros(B, dtype=torch.uint8, device=dev); amp = torch.zeros(B, 2, device=dev); toa = torch.zeros(B, device=dev)
    co = torch.zeros(B, device=dev); ch = torch.zeros(B, 6, 2, device=dev)
    def go():
        t.channel_estimate(xf, off, length, tsc, fl, amp, toa, co, ch, variant52m=True, max_toa=4)
    for _ in range(50): go()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): go()
    e1.record(); torch.cuda.synchronize()
    print('B %6d  channel_estimate alone: %.1f us per launch, detected %.3f' % (B, e0.elapsed_time(e1) / 300 * 1e3,
                                                                              float(((fl & pkg.F_DETECT) != 0).float().mean())))

# the same kernel inside the three-kernel step, float32 and fp16 storage, from the library's own event timers
B = 65536
x, off, length, meta = synth.normal_batch_torch(1, B, tsc, seed=5, device=dev, sigmas=(0.02, 0.1), max_delay=1.0)
xr = torch.view_as_real(x)
q = torch.clamp(torch.round(xr * (2000.0 / float(xr.abs().max()))), -2048, 2048)
for name, smp, fp16 in (("float32", q.contiguous(), False), ("fp16", q.to(torch.float16).contiguous(), True)):
    t = pkg.TrxSig(1, 0); t.use_torch_stream(); t.reserve(B)
    d = dict(flags=torch.zeros(B, dtype=torch.uint8, device=dev), amp=torch.zeros(B, 2, device=dev), toa=torch.zeros(B, device=dev),
             w=torch.zeros(B, 7, 2, device=dev), b=torch.zeros(B, 5, 2, device=dev), soft=torch.zeros(B, 157, device=dev))
    def step():
        t.equalize_normal(smp, off, length, tsc, d['flags'], d['amp'], d['toa'], d['soft'], w=d['w'], b=d['b'], energy_thresh=10.0,
                          variant52m=True, max_toa=4, nsoft=156, soft_stride=157, fp16=fp16)
    for _ in range(50): step()
    torch.cuda.synchronize()
    t.profile_enable(True)
    for _ in range(200): step()
    prof = t.profile_collect(); t.profile_enable(False)
    print('in the step, %s storage: ' % name + ', '.join('%s %.1f us' % (k, v[0] / v[1] * 1e3) for k, v in prof.items()))
