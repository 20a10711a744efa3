"""Config 5 at larger calls: what occupancy alone is worth to the lane-per-burst equaliser kernels (65,536 bursts = 1,024
waves = ONE wave per SIMD).  python tools/config5_sizes.py"""
import json, sys, time, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
import _pkg
pkg = _pkg.load()
import bench_config5
dev = torch.device('cuda:0')
for B in (32768, 65536, 131072, 262144, 524288):
    args = types.SimpleNamespace(bursts=B)
    c = pkg.TrxSig(1, 0); c.use_torch_stream()
    w = bench_config5.Config5(args)
    w.setup(pkg, c, dev, 0, args)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04: w.step()
    torch.cuda.synchronize()
    K = max(30, int(300 * 65536 / B))
    t0 = time.perf_counter()
    for _ in range(K): w.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    c.profile_enable(True)
    for _ in range(30): w.step()
    pf = c.profile_collect(); c.profile_enable(False)
    print(json.dumps({"bursts_per_call": B, "us_per_step": round(dt * 1e6, 1), "Mbursts_per_s": round(B / dt / 1e6, 1),
                      "kernels_us": {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in pf.items()}}), flush=True)
    del w, c
    torch.cuda.empty_cache()
