#!/usr/bin/env python3
"""bench.py -- Mbursts/s of the burst detect+demod hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N>1 without WORLD_SIZE in the environment: this process never touches the GPU; it starts the N ranks itself
  (a child `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py`)
  and exits with the child's code.  Under torchrun (WORLD_SIZE set) it is one of the ranks.

Workload (config.workload): BASELINE config 2 -- 65,536 normal bursts per GPU, 156.25 symbols at
4 samples/symbol (628/624/624/624 complex float32 samples), one training sequence per batch,
synthetic GMSK bursts with random gain, sub-sample delay and AWGN (SNR inf/20/10 dB), resident in
HBM before the timed region.  A "step" is one pass of trxsig_detect_demod_normal_batch over the
batch: energy detect + TSC correlate + peak/valley detect + GMSK demodulation to 148 soft bits.
Multi-GPU: each rank owns an independent batch (weak scaling); the only collective is the
init-time RCCL broadcast of the constant tables.  `value` is whole-job bursts/s over all ranks,
timed between barriers, MAX over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SPS = 4
TSC = 2
BURSTS_PER_GPU = 65536
NSOFT = 148
# algorithmic bytes per burst: read 8*625 + write 4*148 soft + 16 B metadata
ALG_READ = 8 * 625
ALG_WRITE = 4 * NSOFT + 16
ALG_BYTES = ALG_READ + ALG_WRITE
# per-kernel algorithmic bytes per burst (DESIGN.md "Kernels"):
KERNEL_ALG_BYTES = {
    "k_rach_corr": ALG_READ + 1012, "k_rach_peak": 1012 + 17,
    "k_tsc_corr": 8 * 36 * SPS + 8 * 20 * SPS + 8 * 44,     # window + energy window read, record write
    "k_tsc_peak": 8 * 44 + 13 + 4,                          # record read, flags/amp/toa/avgpwr write
    "k_demod": ALG_READ + 13 + 4 * NSOFT,                   # whole burst + amp/toa/flags read, soft write
    "k_normal_fused": ALG_BYTES, "k_normal_chain": ALG_BYTES,
}


def measured_traffic(kernel, bursts):
    """HBM bytes per launch of `kernel` from the committed PMC run (profiles/traffic.json: rocprofv3
    FETCH_SIZE x2 + WRITE_SIZE, separate passes, same bench command), scaled to this batch size."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kernels"][kernel]
        return int(t["hbm_bytes_per_launch"] * (bursts / float(t["bursts_per_launch"])))
    except Exception:
        return None


def cpu_baseline(x_host, off, length, sps, tsc, target_seconds=8.0):
    """The CPU oracle (a port of the reference's algorithm, oracle/sigproc_oracle.c) timed on this
    box's host cores over a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oraclebind
    o = oraclebind.Oracle(sps)
    cores = min(os.cpu_count() or 1, 16)
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    B = len(off)
    t0 = time.perf_counter()
    o.normal_batch(x_host, off[:1024], length[:1024], tsc, nthreads=1)
    t1 = time.perf_counter() - t0
    o.normal_batch(x_host, off, length, tsc, nthreads=cores)            # warm
    t0 = time.perf_counter()
    o.normal_batch(x_host, off, length, tsc, nthreads=cores)
    tp = time.perf_counter() - t0
    reps = max(1, int(target_seconds / max(tp, 1e-3)))
    t0 = time.perf_counter()
    for _ in range(reps):
        o.normal_batch(x_host, off, length, tsc, nthreads=cores)
    tt = time.perf_counter() - t0
    port = {"value": round(B * reps / tt / 1e6, 6), "unit": "Mbursts/s", "cores": cores, "kind": "port",
            "single_thread_Mbursts_per_s": round(1024 / t1 / 1e6, 6),
            "sample": "%d passes over the first %d bursts of the GPU batch (analyzeTrafficBurst + "
                      "demodulateBurst, oracle/sigproc_oracle.c, %d OpenMP threads, %.1f s)" % (reps, B, cores, tt)}
    # the real reference, when its in-place build travelled with the snapshot (oracle/_ref/, built by
    # __graft_entry__.build() where /root/reference exists): timed in a child process that never touches the GPU
    import refbind
    if not refbind.available():
        return port, None
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sample.npz")
        end = int(off[-1] + length[-1])
        np_ = __import__("numpy")
        np_.savez(path, x=x_host[:end], off=off, length=length, sps=sps, tsc=tsc)
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_bench.py"), path, str(cores), "6"],
                               capture_output=True, text=True, timeout=300)
            ref = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:                               # the reference leg is optional; the port leg stands
            sys.stderr.write("reference cpu baseline unavailable: %r\n" % (e,))
            return port, None
    return ref, port


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks as a child
    torch.distributed.run (one process per GPU, as TRXManager runs one transceiver per ARFCN,
    TRXManager/TRXManager.cpp:44-54) and exit with its code.  The parent makes no HIP call (counting devices
    does not initialise the GPU) and never re-execs."""
    import subprocess
    if not args.selftest_cpu:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible -- refusing to run a smaller job under "
                             "that label\n" % (args.gpus, have))
            sys.exit(2)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def selftest_cpu(args):
    """The launch path without a GPU (tests/test_bench_spawn.py, gloo): rendezvous, rank 0 builds the table
    blob, broadcast + checksum on every rank, MAX all-reduce, all-gather of the ranks, one JSON line from
    rank 0.  No burst is processed and no throughput is reported (`value` null)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import _pkg
    pkg = _pkg.load()
    from openbts_ttsou_amd import dist as tdist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0
    if world > 1:
        rank, world = tdist.init_from_env("gloo")
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    blob, _ = tdist.broadcast_tables(pkg, SPS, device=None, src=0)
    lo, hi = tdist.shard_range(args.bursts * world, rank, world)
    tmax = tdist.max_over_ranks(1.0 + rank)
    seen = tdist.ranks_seen(rank, None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": True, "value": None, "n_gpus": world, "ranks_seen": seen, "tmax": tmax,
                          "tables_fnv": int(np.frombuffer(blob.tobytes()[-8:], np.uint64)[0]) if len(blob) >= 8 else 0,
                          "shard0": [lo, hi]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--preheat-ms", type=float, default=40.0,
                    help="untimed passes before the warm-up until this much GPU time has gone by: the device needs "
                         "~15-20 ms of load to leave its idle clocks (measured: 496 Mbursts/s over the first 50 "
                         "steps of a cold run, 573 once it has ramped), whatever --warmup/--steps the caller picks")
    ap.add_argument("--bursts", type=int, default=BURSTS_PER_GPU, help="bursts per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="also check a sample against the CPU oracle")
    ap.add_argument("--workload", choices=["normal", "rach"], default="normal",
                    help="normal = BASELINE config 2 (the headline metric); rach = config 3 (side measurement)")
    ap.add_argument("--chain-lag", type=int, default=None, help="A/B: path 5, tiles between detect and demodulate workgroups")
    ap.add_argument("--path", type=int, default=None, choices=[0, 1, 2, 3, 4, 5],
                    help="A/B: normal-burst implementation (trxsig_set_tuning); default = the library's")
    ap.add_argument("--spec-peak", type=int, default=0, choices=[0, 1, 2],
                    help="A/B, path 0's peak kernel: 0 = two lanes per burst (default), 1 = eight lanes, speculative, 2 = a lane per burst")
    ap.add_argument("--no-fresh", action="store_true", help="skip the rotating-inputs side measurement")
    ap.add_argument("--generic-taps", action="store_true", help="A/B: correlators without the tap-class specialisation")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="exercise the N-rank launch path on the CPU (gloo): rendezvous, table broadcast, reductions; "
                         "no bursts are processed and no number is reported")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args, sys.argv[1:])                     # does not return
    if args.selftest_cpu:
        return selftest_cpu(args)

    import numpy as np
    import torch
    import _pkg
    pkg = _pkg.load()
    from openbts_ttsou_amd import dist as tdist
    from openbts_ttsou_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        rank, world = tdist.init_from_env("nccl")
    else:
        rank, local = 0, 0
        torch.cuda.set_device(0)
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d -- refusing to report one under the other's label\n"
                         % (args.gpus, world))
        sys.exit(2)
    dev = torch.device("cuda", local)

    # constant tables: built once on rank 0, broadcast over RCCL (xGMI), validated, then each rank's
    # context is created from the received device blob
    if world > 1:
        _, tbl = tdist.broadcast_tables(pkg, SPS, device=dev, src=0)
        ctx = pkg.TrxSig(SPS, local, tables_blob=tbl)
    else:
        ctx = pkg.TrxSig(SPS, local)
    ctx.use_torch_stream()
    if args.path is not None:
        ctx.set_tuning(normal_path=args.path)
    if args.chain_lag is not None:
        ctx.set_tuning(chain_lag=args.chain_lag)
    if args.generic_taps:
        ctx.set_tuning(generic_taps=1)
    if args.spec_peak:
        ctx.set_tuning(spec_peak=args.spec_peak)

    B = args.bursts
    rach = args.workload == "rach"
    if rach:
        x, off, length, meta = synth.rach_batch_torch(SPS, B, seed=0xB5E55ED0 + rank, device=dev)
    else:
        x, off, length, meta = synth.normal_batch_torch(SPS, B, TSC, seed=0xB5E55ED0 + rank, device=dev)
    xf = torch.view_as_real(x).contiguous()
    flags = torch.zeros(B, dtype=torch.uint8, device=dev)
    amp = torch.zeros(B, 2, dtype=torch.float32, device=dev)
    toa = torch.zeros(B, dtype=torch.float32, device=dev)
    soft = torch.zeros(B, NSOFT, dtype=torch.float32, device=dev)
    ctx.reserve(B)

    def step():
        if rach:
            ctx.detect_demod_rach(xf, off, length, flags, amp, toa, soft, detect_thresh=5.0, energy_thresh=-1.0,
                                  nsoft=NSOFT, soft_stride=NSOFT)
        else:
            ctx.detect_demod_normal(xf, off, length, TSC, flags, amp, toa, soft, detect_thresh=3.0,
                                    energy_thresh=0.0, nsoft=NSOFT, soft_stride=NSOFT)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # clock ramp (untimed, not part of W): see --preheat-ms
    t_pre = time.perf_counter()
    n_pre = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        n_pre += 20
    for _ in range(args.warmup):
        step()
    # ---- the timed region: exactly K steps between barriers, nothing else on the stream ----
    barrier()
    ctx.timer_start()                       # HIP events on the stream the kernels run on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = tdist.max_over_ranks(elapsed, dev)
    seen = tdist.ranks_seen(rank, dev)
    # ---- per-kernel durations: the same K steps again with every launch bracketed by HIP events
    #      (kept out of the timed region: the extra event records stretch the gaps between kernels)
    ctx.profile_enable(True)
    for _ in range(min(args.steps, 200)):
        step()
    prof = ctx.profile_collect()
    ctx.profile_enable(False)

    # ---- side measurement (outside the timed region): the same steps over THREE different input batches in
    #      rotation (1 GB > the 256 MB memory-side cache), i.e. without the part of the input that a repeated
    #      batch still finds in that cache from the step before
    fresh = None
    if world == 1 and not rach and not args.no_fresh:
        xs = [xf]
        for k in (1, 2):
            xk, offk, lenk, _ = synth.normal_batch_torch(SPS, B, TSC, seed=0xB5E55ED0 + 1000 * k, device=dev)
            assert torch.equal(offk, off) and torch.equal(lenk, length)
            xs.append(torch.view_as_real(xk).contiguous())
        so2 = torch.zeros_like(soft); fl2 = torch.zeros_like(flags); am2 = torch.zeros_like(amp); to2 = torch.zeros_like(toa)
        def step_k(i):
            ctx.detect_demod_normal(xs[i % 3], off, length, TSC, fl2, am2, to2, so2, detect_thresh=3.0,
                                    energy_thresh=0.0, nsoft=NSOFT, soft_stride=NSOFT)
        kf = min(args.steps, 600)
        for i in range(60):
            step_k(i)
        torch.cuda.synchronize()
        tf = time.perf_counter()
        for i in range(kf):
            step_k(i)
        torch.cuda.synchronize()
        tf = time.perf_counter() - tf
        fresh = {"value": round(B * kf / tf / 1e6, 3), "unit": "Mbursts/s", "inputs_in_rotation": 3, "steps": kf}
        del xs, so2

    # results sanity (outside the timed region): detections and hard bits of the clean bursts
    det = (flags & pkg.F_DETECT) != 0
    clean = det & (meta["sigma"] <= 0.1)
    cols = slice(8, 85) if rach else slice(0, 148)
    hard_ok = bool(((soft[clean][:, cols] > 0.5).to(torch.uint8) == meta["bits"][clean][:, cols]).all().item())
    det_frac = float(det.float().mean().item())

    if rank != 0:
        return
    value = world * B * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    dom = max(prof.items(), key=lambda kv: kv[1][0]) if prof else (None, (0.0, 0))
    roof = None
    if dom[0]:
        avg_ms = dom[1][0] / max(dom[1][1], 1)
        achieved = KERNEL_ALG_BYTES.get(dom[0], ALG_BYTES) * B / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(dom[0], B),
                "alg_bytes_per_launch": KERNEL_ALG_BYTES.get(dom[0], ALG_BYTES) * B,
                "avg_kernel_ms": round(avg_ms, 4),
                "alg_bytes_per_burst": KERNEL_ALG_BYTES.get(dom[0], ALG_BYTES),
                "pipeline_achieved": round(ALG_BYTES * B * args.steps / (ev_ms * 1e-3) / 1e9, 1),
                "pipeline_frac": round(ALG_BYTES * B * args.steps / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "kernels_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}}
    out = {
        "metric": "Mbursts/s (156.25-sym @ 4 sps) demod+detect", "value": round(value, 3), "unit": "Mbursts/s",
        "n_gpus": world, "ranks_seen": seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("config3: %d access bursts/GPU, sps=4, detectRACHBurst over all lags (thr 5.0) + demod "
                                "to %d soft bits" % (B, NSOFT)) if rach else
                               ("config2: %d normal bursts/GPU, sps=4, 628/624/624/624 complex f32 samples, "
                                "TSC %d, detect (thr 3.0) + demod to %d soft bits" % (B, TSC, NSOFT)),
                   "bursts_per_gpu": B, "sps": SPS, "parallelism": "burst-sharded x%d (no data-path collective)" % world},
        "hip_event_ms_per_step": round(ev_ms / args.steps, 4), "preheat_steps": n_pre,
        "detected_frac": round(det_frac, 4), "clean_hard_bits_ok": hard_ok,
        "roofline": roof, "fresh_inputs": fresh,
    }
    if not args.no_cpu_baseline and world == 1 and not rach:
        n = min(B, 16384)
        end = int(off[n - 1].item() + length[n - 1].item())
        xh = x[:end].cpu().numpy()
        main_leg, port_leg = cpu_baseline(xh, off[:n].cpu().numpy(), length[:n].cpu().numpy(), SPS, TSC)
        out["cpu_baseline"] = main_leg
        if port_leg is not None:
            out["cpu_port"] = port_leg                   # the oracle port beside the real reference
        if args.check:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oraclebind
            ok, oamp, otoa, osoft = oraclebind.Oracle(SPS).normal_batch(xh, off[:n].cpu().numpy(),
                                                                        length[:n].cpu().numpy(), TSC, nthreads=8)
            same = (np.array_equal(det[:n].cpu().numpy(), ok.astype(bool)) and
                    np.array_equal(soft[:n].cpu().numpy(), osoft) and np.array_equal(toa[:n].cpu().numpy(), otoa))
            out["oracle_check_first_%d" % n] = bool(same)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
