import os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, _pkg
pkg=_pkg.load()
from openbts_ttsou_amd import synth
dev=torch.device('cuda:0'); B=65536
x,off,length,meta=synth.normal_batch_torch(4,B,2,seed=1,device=dev)
xf=torch.view_as_real(x).contiguous()
c=pkg.TrxSig(4,0); c.use_torch_stream(); c.reserve(B)
f=torch.zeros(B,dtype=torch.uint8,device=dev); a=torch.zeros(B,2,device=dev); t=torch.zeros(B,device=dev); s=torch.zeros(B,148,device=dev)
def step(): c.detect_demod_normal(xf,off,length,2,f,a,t,s,nsoft=148,soft_stride=148)
for _ in range(300): step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(100): step()
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print('host submit %.1f us/step; device drain after submit %.1f us/step equivalent' % ((t1-t0)*1e4, (t2-t0)*1e4))
