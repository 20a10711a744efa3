import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import torch, numpy as np
import _pkg
pkg=_pkg.load()
t=pkg.TrxSig(4,0); t.use_torch_stream()
B=65536; nb=B//4
soft=torch.rand(B,148,device='cuda')
frames=torch.zeros(nb,23,dtype=torch.uint8,device='cuda'); ok=torch.zeros(nb,dtype=torch.uint8,device='cuda')
o3=[torch.zeros(B,dtype=torch.uint8,device='cuda') for _ in range(3)]
for name,fn in (("xcch",lambda: t.fec_xcch_decode(soft,nb,frames,ok,wire=True)),("rach",lambda: t.fec_rach_decode(soft,B,o3[0],o3[1],o3[2],wire=True))):
    for _ in range(300): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    K=300
    for _ in range(K): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/K
    print(name,"%.1f us per 65536 bursts  -> %.1f Mbursts/s"%(dt*1e6,B/dt/1e6))
import fecbind
o=fecbind.FecOracle()
s=soft[:16384].cpu().numpy()
t0=time.perf_counter(); o.xcch_decode_batch(s,wire=True,nthreads=16); dt=time.perf_counter()-t0
print("cpu xcch 16 threads: %.3f Mbursts/s"%(16384/dt/1e6))
