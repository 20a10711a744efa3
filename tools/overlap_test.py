import sys, time
sys.path.insert(0,'.')
import torch, numpy as np
import _pkg
pkg=_pkg.load()
from openbts_ttsou_amd import synth
dev=torch.device('cuda:0')
B=65536
x,off,length,meta=synth.normal_batch_torch(4,B,2,seed=1,device=dev)
xf=torch.view_as_real(x).contiguous()
for nstream in (1,2,4,8):
    ctxs=[]; streams=[]
    for i in range(nstream):
        c=pkg.TrxSig(4,0); s=torch.cuda.Stream(); c.set_stream(s.cuda_stream); c.reserve(B); ctxs.append(c); streams.append(s)
    h=B//nstream
    flags=torch.zeros(B,dtype=torch.uint8,device=dev); amp=torch.zeros(B,2,device=dev); toa=torch.zeros(B,device=dev); soft=torch.zeros(B,148,device=dev)
    # per-chunk views; offsets are absolute so samples base pointer stays
    def step():
        for i,c in enumerate(ctxs):
            sl=slice(i*h,(i+1)*h)
            c.detect_demod_normal(xf, off[sl], length[sl], 2, flags[sl], amp[sl], toa[sl], soft[sl], nsoft=148, soft_stride=148)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    K=50
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/K
    print(nstream,'streams: %.1f us/step  %.1f Mbursts/s'%(dt*1e6, B/dt/1e6))
