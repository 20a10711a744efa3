#!/usr/bin/env python3
"""Side measurement for the L1 FEC row: k_fec_viterbi on the soft bits of 65,536 bursts
resident in HBM -- 16,384 XCCH blocks, then 65,536 RACH bursts -- with the CPU oracle timed beside it.
Prints one JSON line per workload in the shape of bench.py's.  Run on the GPU box: python tools/fec_bench.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import _pkg

pkg = _pkg.load()
t = pkg.TrxSig(4, 0); t.use_torch_stream()
B = 65536; nb = B // 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
soft = torch.rand(B, 148, device="cuda", generator=g)
frames = torch.zeros(nb, 23, dtype=torch.uint8, device="cuda"); ok = torch.zeros(nb, dtype=torch.uint8, device="cuda")
o3 = [torch.zeros(B, dtype=torch.uint8, device="cuda") for _ in range(3)]
import fecbind
o = fecbind.FecOracle()
cores = min(os.cpu_count() or 1, 16)
work = {
    "xcch": (lambda: t.fec_xcch_decode(soft, nb, frames, ok, wire=True), 4 * 114 * 4 * nb + 24 * nb,
             lambda s: o.xcch_decode_batch(s, wire=True, nthreads=cores)),
    "rach": (lambda: t.fec_rach_decode(soft, B, o3[0], o3[1], o3[2], wire=True), (36 * 4 + 3) * B,
             lambda s: o.rach_decode_batch(s, wire=True, nthreads=cores)),
}
for name, (fn, alg_bytes, cpu) in work.items():
    for _ in range(400): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 500
    for _ in range(K): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    t.profile_enable(True)
    for _ in range(50): fn()
    prof = t.profile_collect(); t.profile_enable(False)
    kms = prof["k_fec_viterbi"][0] / prof["k_fec_viterbi"][1]
    s = soft[:16384].cpu().numpy()
    c0 = time.perf_counter(); reps = 0
    while time.perf_counter() - c0 < 4.0:
        cpu(s); reps += 1
    cdt = (time.perf_counter() - c0) / reps
    print(json.dumps({
        "metric": "Mbursts/s through the L1 FEC soft decode (%s)" % name, "value": round(B / dt / 1e6, 2), "unit": "Mbursts/s",
        "n_gpus": 1, "steps": K, "ms_per_step": round(dt * 1e3, 4), "dtype": "f32 metrics / u32 paths", "data": "synthetic",
        "config": {"workload": "%s: soft bits of %d bursts (uniform random in [0,1]) resident in HBM, UDP-hop quantisation on"
                               % (name, B)},
        "roofline": {"bound": "hbm", "kernel": "k_fec_viterbi", "achieved": round(alg_bytes / (kms * 1e-3) / 1e9, 1),
                     "peak": 8000.0, "unit": "GB/s", "frac": round(alg_bytes / (kms * 1e-3) / 1e9 / 8000.0, 4),
                     "avg_kernel_ms": round(kms, 4), "traffic": None,
                     "note": "latency/VALU bound: a ~50-instruction dependent chain per trellis step, 4 waves per SIMD"},
        "cpu_baseline": {"value": round(16384 / cdt / 1e6, 3), "unit": "Mbursts/s", "cores": cores, "kind": "port",
                         "sample": "%d passes over the first 16384 bursts (oracle/fec_oracle.c, OpenMP)" % reps}}), flush=True)
