#!/usr/bin/env python3
"""bench.py -- Mbursts/s of the burst detect+demod hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload normal|rach|config4|config5]
  N>1 without WORLD_SIZE in the environment: this process never touches the GPU; it starts the N ranks itself
  (a child `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py`)
  and exits with the child's code.  Under torchrun (WORLD_SIZE set) it is one of the ranks.

Workloads (config.workload; the default, `normal`, is the headline):
  normal   BASELINE config 2 -- 65,536 normal bursts per GPU, 156.25 symbols at 4 samples/symbol (628/624/624/624 complex
           float32 samples), one training sequence per batch, synthetic GMSK bursts with random gain, sub-sample delay and
           AWGN (SNR inf/20/10 dB), resident in HBM before the timed region.  A step = one pass of
           trxsig_detect_demod_normal_batch: energy detect + TSC correlate + peak/valley detect + demodulation to 148 soft bits.
  rach     config 3 -- 65,536 access bursts per GPU, detectRACHBurst over every lag + demodulation.
  config4  S ARFCN streams per GPU of int16 I/Q at 400 kS/s, K chunks of 864 samples per step, through the Transceiver group on
           the fused receive front end (trxsig_trxgroup_pull_rxfe): unUSRPify + polyphase resample 65*sps:96 behind the 192-sample
           history + 157/156/156/156 slicing + per-slot expectedCorrType (every eighth ARFCN carries channel combination V on
           timeslot 0, i.e. access-burst slots; all other slots combination I) + midamble / access-burst detection + the
           per-ARFCN energy-threshold state machine + demodulation, the detectors computing their samples from the int16 chunks.
           --stateless-frontend: the same without schedule or state (trxsig_rxfe_push_detect_demod_normal, TSC on every slot,
           fixed thresholds); --unfused-frontend: through the resampled stream (push + pop + detect).
           (BASELINE's "8 ARFCN, one per GPU" run is --gpus 8: a stream set per rank.)
  config5  the Transceiver52M receive leg at one sample per symbol, fp16 sample storage: energy gate, windowed midamble
           correlation with channel estimate, designDFE (Nf = 7), equalizeBurst.
Multi-GPU: each rank owns an independent batch / stream set (weak scaling, one engine per ARFCN set as
TRXManager/TRXManager.cpp:44-54); the only collective is the init-time RCCL broadcast of the constant tables.  `value` is
whole-job bursts/s over all ranks, timed between barriers, MAX over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NSOFT = 148


def measured_traffic(kernel, units, key="bursts_per_launch"):
    """HBM bytes per launch of `kernel` from the committed PMC run (profiles/traffic.json: rocprofv3 FETCH_SIZE x2 +
    WRITE_SIZE, separate passes, same bench command), scaled to this launch size."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kernels"][kernel]
        return int(t["hbm_bytes_per_launch"] * (units / float(t[key])))
    except Exception:
        return None


def host_cores():
    """Worker count for the CPU baseline: every core this process may run on (scheduler affinity, then the cgroup's CPU quota
    where one is set -- a GPU box hands a one-GPU job its share of a larger machine), at most 64 so that the default run stays
    within the box's process budget."""
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(cores, 64))


def host_info():
    """What the box is, beside `cores` (the threads actually used): logical CPUs of the machine, CPUs this process may use, model."""
    info = {"host_logical_cpus": os.cpu_count() or 0}
    try:
        info["host_cpus_allowed"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return info


# ======================================================================================================================
class Normal:
    """BASELINE config 2 (the headline) and, with rach=True, config 3."""
    sps, tsc = 4, 2
    dtype = "f32"

    def __init__(self, args, rach=False):
        self.rach = rach
        self.soft_mode = getattr(args, "soft_mode", "tolerance")
        self.B = args.bursts or 65536
        self.alg_read = 8 * 625                               # 8*N-bar bytes read per burst
        self.alg_bytes = self.alg_read + 4 * NSOFT + 16       # + 148 soft bits + flag / amp / TOA
        s = self.sps
        self.kernel_alg = {                                   # per-kernel algorithmic bytes per burst (DESIGN.md "Kernels")
            "k_tsc_corr": 8 * 36 * s + 8 * 20 * s + 8 * 44,   # window + energy window read, record write
            "k_tsc_peak": 8 * 44 + 13 + 4,                    # record read, flags / amp / toa / avgpwr write
            "k_demod": self.alg_read + 13 + 4 * NSOFT,        # whole burst + amp / toa / flags read, soft write
            "k_normal_fused": self.alg_bytes, "k_normal_chain": self.alg_bytes,
            "k_rach_corr": self.alg_read + 8 * 25 + 16 + 17,  # k_rach_front: whole burst read, 25-slot record + 4 floats written
            "k_rach_peak": 8 * 25 + 16 + 17,
        }
        self.kernel_names = {"k_rach_corr": "k_rach_front", "k_rach_peak": "k_rach_peak2+k_rach_fast(list)", "k_tsc_peak": "k_tsc_peak2"}

    def setup(self, pkg, ctx, dev, rank, args):
        import torch
        from openbts_ttsou_amd import synth
        self.pkg, self.ctx, self.dev, self.torch, self.synth = pkg, ctx, dev, torch, synth
        B = self.B
        if self.rach:
            x, off, length, meta = synth.rach_batch_torch(self.sps, B, seed=0xB5E55ED0 + rank, device=dev)
        else:
            x, off, length, meta = synth.normal_batch_torch(self.sps, B, self.tsc, seed=0xB5E55ED0 + rank, device=dev)
        self.x, self.off, self.length, self.meta = x, off, length, meta
        self.xf = torch.view_as_real(x).contiguous()
        self.flags = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.amp = torch.zeros(B, 2, dtype=torch.float32, device=dev)
        self.toa = torch.zeros(B, dtype=torch.float32, device=dev)
        self.soft = torch.zeros(B, NSOFT, dtype=torch.float32, device=dev)
        ctx.set_soft_mode(pkg.SOFT_TOLERANCE if self.soft_mode == "tolerance" else pkg.SOFT_EXACT)
        ctx.reserve(B)

    def step(self):
        if self.rach:
            self.ctx.detect_demod_rach(self.xf, self.off, self.length, self.flags, self.amp, self.toa, self.soft, detect_thresh=5.0,
                                       energy_thresh=-1.0, nsoft=NSOFT, soft_stride=NSOFT)
        else:
            self.ctx.detect_demod_normal(self.xf, self.off, self.length, self.tsc, self.flags, self.amp, self.toa, self.soft,
                                         detect_thresh=3.0, energy_thresh=0.0, nsoft=NSOFT, soft_stride=NSOFT)

    def units_per_step(self):
        return self.B

    def describe(self, world):
        w = ("config3: %d access bursts/GPU, sps=4, detectRACHBurst over all lags (thr 5.0) + demod to %d soft bits" % (self.B, NSOFT)) \
            if self.rach else ("config2: %d normal bursts/GPU, sps=4, 628/624/624/624 complex f32 samples, TSC %d, detect (thr 3.0) + "
                               "demod to %d soft bits" % (self.B, self.tsc, NSOFT))
        w += ("; soft mode TOLERANCE (trxsig_set_soft_mode: flags / amp / TOA / hard bits bit-exact, soft bits within 7.4e-5 of the "
              "reference's -- north_star's 1e-4)" if self.soft_mode == "tolerance" else
              "; soft mode EXACT (every soft bit IEEE-equal to the reference's)")
        return {"workload": w, "bursts_per_gpu": self.B, "sps": self.sps, "soft_mode": self.soft_mode,
                "parallelism": "burst-sharded x%d (no data-path collective)" % world}

    def other_soft_mode(self, steps):
        """Side measurement (never `value`): the same K steps in the OTHER soft mode, and the two modes' outputs against each other
        on the whole batch: flags / amp / TOA / hard bits must be identical, the soft bits' largest difference is reported."""
        torch, pkg = self.torch, self.pkg
        B = self.B
        hard_a = torch.zeros(B, NSOFT, dtype=torch.uint8, device=self.dev)
        hard_b = torch.zeros_like(hard_a)

        def run(hard):
            if self.rach:
                self.ctx.detect_demod_rach(self.xf, self.off, self.length, self.flags, self.amp, self.toa, self.soft, hard=hard, detect_thresh=5.0,
                                           energy_thresh=-1.0, nsoft=NSOFT, soft_stride=NSOFT)
            else:
                self.ctx.detect_demod_normal(self.xf, self.off, self.length, self.tsc, self.flags, self.amp, self.toa, self.soft, hard=hard,
                                             detect_thresh=3.0, energy_thresh=0.0, nsoft=NSOFT, soft_stride=NSOFT)
            self.ctx.synchronize(); torch.cuda.synchronize()
            return self.flags.clone(), self.amp.clone(), self.toa.clone(), self.soft.clone()
        mine = run(hard_a)
        other = "exact" if self.soft_mode == "tolerance" else "tolerance"
        self.ctx.set_soft_mode(pkg.SOFT_EXACT if other == "exact" else pkg.SOFT_TOLERANCE)
        theirs = run(hard_b)
        for _ in range(max(steps // 10, 5)):
            self.step()
        self.ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.ctx.synchronize(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.ctx.profile_enable(True)
        for _ in range(min(steps, 100)):
            self.step()
        pf = self.ctx.profile_collect()
        self.ctx.profile_enable(False)
        self.ctx.set_soft_mode(pkg.SOFT_TOLERANCE if self.soft_mode == "tolerance" else pkg.SOFT_EXACT)
        self.step(); self.ctx.synchronize(); torch.cuda.synchronize()      # (the outputs the later checks read are this mode's again)
        d = (mine[3].double() - theirs[3].double()).abs()
        # what goes on the wire: (char) round(soft * 255) (Transceiver.cpp:669)
        wa, wb = torch.round(mine[3] * 255.0).to(torch.int32), torch.round(theirs[3] * 255.0).to(torch.int32)
        wire_diff = (wa != wb)
        return {"soft_mode": other, "value": round(B * steps / dt / 1e6, 3), "unit": "Mbursts/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
                "kernels_ms": {self.kernel_names.get(k, k): round(v[0] / max(v[1], 1), 4) for k, v in pf.items()},
                "against_this_mode": {"flags_identical": bool(torch.equal(mine[0], theirs[0])), "amp_identical": bool(torch.equal(mine[1], theirs[1])),
                                      "toa_identical": bool(torch.equal(mine[2], theirs[2])), "hard_bits_identical": bool(torch.equal(hard_a, hard_b)),
                                      "soft_max_abs_diff": float(d.max().item()), "soft_values_not_identical": round(float((d > 0).float().mean().item()), 4),
                                      "guaranteed_bound": 7.4e-5,
                                      "wire_bytes_differ": round(float(wire_diff.float().mean().item()), 7),
                                      "wire_bytes_max_step": int((wa - wb).abs().max().item())}}

    def sanity(self):
        torch = self.torch
        det = (self.flags & self.pkg.F_DETECT) != 0
        clean = det & (self.meta["sigma"] <= 0.1)
        cols = slice(8, 85) if self.rach else slice(0, 148)
        hard_ok = bool(((self.soft[clean][:, cols] > 0.5).to(torch.uint8) == self.meta["bits"][clean][:, cols]).all().item())
        return {"detected_frac": round(float(det.float().mean().item()), 4), "clean_hard_bits_ok": hard_ok}

    pipelined_key = "lever_demod_beside_next_correlator"

    def pipelined(self, steps):
        """Side measurement (never `value`): TRXSIG_TUNE_DEMOD_BESIDE -- a step returns with its demodulator running on the
        context's side stream, so the NEXT step's correlator (VALU-bound) runs beside it (HBM-bound).  The lever DESIGN 6 priced
        for "a launch that demodulates batch i while correlating batch i+1", without a new kernel.  Same soft bits (checked here)."""
        if self.rach:
            return None
        torch = self.torch
        self.step(); self.ctx.synchronize()
        ref_soft, ref_flags = self.soft.clone(), self.flags.clone()
        self.ctx.set_tuning(demod_beside=1)
        for _ in range(max(steps // 10, 5)):
            self.step()
        self.ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.ctx.synchronize(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = bool(torch.equal(self.soft, ref_soft) and torch.equal(self.flags, ref_flags))
        self.ctx.set_tuning(demod_beside=0)
        return {"value": round(self.B * steps / dt / 1e6, 3), "unit": "Mbursts/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
                "same_outputs_as_default": same,
                "what": "trxsig_set_tuning(TRXSIG_TUNE_DEMOD_BESIDE, 1): the demodulator of step i runs on a side stream beside the "
                        "correlator of step i+1; d_soft (and the reads of the samples) complete behind trxsig_synchronize"}

    def fresh_inputs(self, steps):
        """Side measurement (outside the timed region): the same steps over THREE different input batches in rotation
        (1 GB > the 256 MB memory-side cache), i.e. without the part of the input a repeated batch still finds in that cache."""
        if self.rach:
            return None
        torch, synth = self.torch, self.synth
        xs = [self.xf]
        for k in (1, 2):
            xk, offk, lenk, _ = synth.normal_batch_torch(self.sps, self.B, self.tsc, seed=0xB5E55ED0 + 1000 * k, device=self.dev)
            assert torch.equal(offk, self.off) and torch.equal(lenk, self.length)
            xs.append(torch.view_as_real(xk).contiguous())
        so2 = torch.zeros_like(self.soft); fl2 = torch.zeros_like(self.flags); am2 = torch.zeros_like(self.amp); to2 = torch.zeros_like(self.toa)

        def step_k(i):
            self.ctx.detect_demod_normal(xs[i % 3], self.off, self.length, self.tsc, fl2, am2, to2, so2, detect_thresh=3.0,
                                         energy_thresh=0.0, nsoft=NSOFT, soft_stride=NSOFT)
        kf = min(steps, 600)
        for i in range(60):
            step_k(i)
        torch.cuda.synchronize()
        tf = time.perf_counter()
        for i in range(kf):
            step_k(i)
        torch.cuda.synchronize()
        tf = time.perf_counter() - tf
        self.ctx.profile_enable(True)                       # the same rotation with every launch bracketed by HIP events
        for i in range(min(kf, 150)):
            step_k(i)
        pf = self.ctx.profile_collect()
        self.ctx.profile_enable(False)
        return {"value": round(self.B * kf / tf / 1e6, 3), "unit": "Mbursts/s", "inputs_in_rotation": 3, "steps": kf, "prof": pf}

    def cpu_baseline(self, check):
        """The real reference (oracle/_ref, when its in-place build travelled with the snapshot) and the oracle port on this
        box's host cores over a bounded sample of the same workload; optionally the first bursts value-exact vs the oracle."""
        np_ = __import__("numpy")
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oraclebind
        n = min(self.B, 16384 if not self.rach else 4096)
        off = self.off[:n].cpu().numpy(); length = self.length[:n].cpu().numpy()
        xh = self.x[:int(off[-1] + length[-1])].cpu().numpy()
        o = oraclebind.Oracle(self.sps)
        cores = host_cores()
        run = (lambda nt, cnt=n: o.rach_batch(xh, off[:cnt], length[:cnt], nthreads=nt)) if self.rach else \
              (lambda nt, cnt=n: o.normal_batch(xh, off[:cnt], length[:cnt], self.tsc, nthreads=nt))
        n1 = min(n, 1024 if not self.rach else 256)
        t0 = time.perf_counter(); run(1, n1); t1 = time.perf_counter() - t0
        run(cores)
        t0 = time.perf_counter(); run(cores); tp = time.perf_counter() - t0
        reps = max(1, int(8.0 / max(tp, 1e-3)))
        t0 = time.perf_counter()
        for _ in range(reps):
            res = run(cores)
        tt = time.perf_counter() - t0
        what = "detectRACHBurst + demodulateBurst" if self.rach else "analyzeTrafficBurst + demodulateBurst"
        port = {"value": round(n * reps / tt / 1e6, 6), "unit": "Mbursts/s", "cores": cores, "kind": "port",
                "single_thread_Mbursts_per_s": round(n1 / t1 / 1e6, 6),
                "sample": "%d passes over the first %d bursts of the GPU batch (%s, oracle/sigproc_oracle.c, %d OpenMP threads, "
                          "%.1f s)" % (reps, n, what, cores, tt)}
        out = {"cpu_baseline": port}
        if check:
            ok, oamp, otoa, osoft = res
            det = ((self.flags[:n] & self.pkg.F_DETECT) != 0).cpu().numpy()
            gsoft = self.soft[:n].cpu().numpy()
            if self.soft_mode == "tolerance":              # hard bits identical, soft bits within the guaranteed bound
                soft_ok = bool(np_.array_equal(gsoft > 0.5, osoft[:, :NSOFT] > 0.5) and
                               np_.abs(gsoft.astype(np_.float64) - osoft[:, :NSOFT]).max() <= 7.4e-5)
                out["oracle_check_soft_max_abs_err"] = float(np_.abs(gsoft.astype(np_.float64) - osoft[:, :NSOFT]).max())
            else:
                soft_ok = bool(np_.array_equal(gsoft, osoft))
            same = (np_.array_equal(det, ok.astype(bool)) and soft_ok and np_.array_equal(self.toa[:n].cpu().numpy(), otoa))
            out["oracle_check_first_%d" % n] = bool(same)
        import refbind
        if not refbind.available():
            return out
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "sample.npz")
            np_.savez(path, x=xh, off=off, length=length, sps=self.sps, tsc=self.tsc, kind="rach" if self.rach else "normal")
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_bench.py"), path, str(cores), "6"],
                                   capture_output=True, text=True, timeout=300)
                out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
                out["cpu_port"] = port                      # the oracle port beside the real reference
            except Exception as e:                          # the reference leg is optional; the port leg stands
                sys.stderr.write("reference cpu baseline unavailable: %r\n" % (e,))
        return out


# ======================================================================================================================
class Config4:
    """S int16 I/Q streams per GPU -> rxfe push (convert + resample) -> pop (slice) -> normal-burst detect + demod."""
    sps, tsc = 4, 2
    dtype = "int16 in, f32 arithmetic"

    def __init__(self, args):
        self.S = args.streams
        self.K = args.chunks
        # --reference-chain: the reference's OWN configuration -- one sample per symbol, RadioInterface's 65 : 96 resampler with
        # its createLPF(., 961, 65) table, Transceiver with the equaliser (designDFE + equalizeBurst behind the per-slot channel
        # cache) -- through the Transceiver group (equalising leg) on a push / pop front end
        self.refchain = bool(getattr(args, "reference_chain", False))
        self.soft_mode = getattr(args, "soft_mode", "tolerance")
        if self.refchain:
            self.sps = 1
        self.per_chunk = 585 * self.sps                      # resampled samples per 864-sample chunk
        self.alg_bytes = 4 * 625 * 96 // (65 * self.sps) + 4 * NSOFT + 16   # int16 in, soft bits out
        self.kernel_alg = {"k_resample": None}               # per stream-chunk, see roofline()
        self.kernel_names = {"k_tsc_peak": "k_tsc_peak2", "k_resample": "k_rx_resample"}
        # default: the Transceiver group on the fused front end (schedule + per-ARFCN state machine);
        # --stateless-frontend: trxsig_rxfe_push_detect_demod_normal (no schedule, no state; the detectors compute their samples
        # from the int16 chunks, no resampled stream in HBM);
        # --unfused-frontend: push + pop + trxsig_detect_demod_normal_batch through the complex float32 stream
        # --wideband C: the channeliser -- S / C wideband streams at 8 x 400 kS/s with C carriers each, mixed down and resampled per
        # carrier in one kernel into the receive buffers, then pop + the normal-burst detector (through the resampled stream)
        self.wide = int(getattr(args, "wideband", 0) or 0)
        self.fused = not bool(getattr(args, "unfused_frontend", False)) and not self.wide and not self.refchain
        self.group = (self.fused and not bool(getattr(args, "stateless_frontend", False))) or self.refchain
        if self.wide:
            self.kernel_names["k_resample"] = "k_resample<int16 wideband, mix>"
        if self.group:
            self.beside_kernels = ("k_group_replay",)      # the state machine: a latency chain of a few workgroups, not priced against HBM
            self.kernel_names.update({"k_rach_corr": "k_rach_front_rx", "k_rach_peak": "k_rach_peak2+k_rach_fast_rx(list)"} if self.fused else
                                     {"k_rach_corr": "k_rach_front", "k_rach_peak": "k_rach_peak2+k_rach_fast(list)",
                                      "k_eq_dfe": "k_eq_dfe2" if getattr(args, "eq_tail", 1) == 2 else "k_eq_dfe4",
                                      "k_eq_detect": "k_eq_list+k_eq_estimate_wave"})
            self.kernel_alg.update({"k_rach_corr": 4 * 236 + 8 * 25 + 16 + 17, "k_rach_peak": 8 * 25 + 16 + 17, "k_group_replay": 16 + 4 + 1 + 8})
        if self.fused:
            self.kernel_names.update({"k_demod": "k_demod_rx", "k_tsc_corr": "k_tsc_corr_rx"})
            # per burst: the raw stretch behind the burst (236 int16 pairs) / behind its two windows (140) read, soft bits / record written
            self.kernel_alg.update({"k_demod": 4 * 236 + 13 + 4 * NSOFT, "k_tsc_corr": 4 * 140 + 8 * 44, "k_tsc_peak": 8 * 44 + 13 + 4})

    def setup(self, pkg, ctx, dev, rank, args):
        import numpy as np
        import torch
        from openbts_ttsou_amd import synth
        from openbts_ttsou_amd.frontend import RxFrontEnd
        self.pkg, self.ctx, self.dev, self.torch = pkg, ctx, dev, torch
        S, K, sps = self.S, self.K, self.sps
        # synthetic radio streams: back-to-back normal bursts (157-156-156-156 symbols) modulated on the device, brought to
        # 400 kS/s by linear interpolation (the bench needs realistic, detectable content, not a calibrated radio), int16
        # The input must stay aligned with the front end's 157-156-156-156 schedule from step to step: 125 chunks are exactly
        # 117 groups, so the generated stream is KT = a multiple of 125 chunks long and consecutive steps push consecutive
        # K-chunk segments of it, wrapping where the stream is aligned again.
        KT = K * 125 // __import__("math").gcd(K, 125)
        self.KT, self.seg = KT, 0
        K = KT
        nb = (K * 585 // 156 + 4 + 3) // 4 * 4               # whole 157-156-156-156 groups: every stream the same length
        gen = torch.Generator(device=dev); gen.manual_seed(0xC0F14 + rank)
        x, off, length, meta = synth.normal_batch_torch(sps, S * nb, self.tsc, seed=0xC0F14 + rank, device=dev, sigmas=(0.02, 0.05))
        hi = x.reshape(-1)[: S * (x.numel() // S)].reshape(S, -1)
        n_lo = K * 864
        t = torch.arange(n_lo, device=dev, dtype=torch.float64) * (65.0 * sps / 96.0)
        i0 = t.floor().long().clamp(max=hi.shape[1] - 2); fr = (t - i0).to(torch.float32)
        lo = hi[:, i0] * (1 - fr) + hi[:, i0 + 1] * fr
        lo = lo * (8000.0 / lo.abs().amax(dim=1, keepdim=True))
        iq = torch.stack([lo.imag, lo.real], dim=2).round().clamp(-32768, 32767).to(torch.int16).contiguous()   # Q first (I/Q flipped)
        K = self.K
        self.iq = iq                                          # [S, KT*864, 2]; a step pushes iq[:, seg*K*864 : (seg+1)*K*864]
        self.segs = [iq[:, i * K * 864:(i + 1) * K * 864].contiguous() for i in range(KT // K)]
        # createLPF(cutoff, 961, 65*sps) as pullBuffer asks for it -- designed for THIS ratio (synth.design_lpf says why the
        # reference's fixed table, made for 65:96, is not used at sps 4); the taps are an argument of the library
        self.lpf = synth.design_lpf(961, 65 * sps)
        if self.refchain:                                     # radioInterface.cpp:230-234 at sps 1: the reference's table (a data fixture)
            raw = np.load(os.path.join(ROOT, "tests", "golden", "resample.npz"))["sendLPF_961_raw"]
            h = pkg.TrxHost(sps, dev.index or 0)
            self.lpf = h.create_lpf(raw, 65.0)
            h.close()
        if self.wide:
            # carriers 400 kHz apart round the centre of a 3.2 MS/s stream; each narrowband stream is brought to the wideband rate
            # by linear interpolation, shifted to its carrier and summed (content for a throughput run, not a calibrated radio)
            C, CW = self.wide, 8
            assert S % C == 0 and C <= 8
            offs = (torch.arange(C, device=dev, dtype=torch.float64) - (C - 1) / 2.0) * 400e3
            self.freqs = (-2.0 * np.pi * offs / (400e3 * CW)).to(torch.float32).cpu().numpy()
            self.lpf = synth.design_lpf(8001, 65 * sps, beta=6.0, cutoff=0.09)
            nw = KT * 864 * CW
            tw = torch.arange(nw, device=dev, dtype=torch.float64) / CW
            i0 = tw.floor().long().clamp(max=KT * 864 - 2); fr = (tw - i0).to(torch.float32)
            lo_c = (iq[:, :, 1].to(torch.float32) + 1j * iq[:, :, 0].to(torch.float32))          # [S, KT*864] complex
            wide = torch.zeros(S // C, nw, dtype=torch.complex64, device=dev)
            ph = torch.arange(nw, device=dev, dtype=torch.float64)
            for k in range(C):
                rot = torch.exp(1j * (2.0 * np.pi * float(offs[k]) / (400e3 * CW)) * ph).to(torch.complex64)
                xk = lo_c[k::C]
                wide += (xk[:, i0] * (1 - fr) + xk[:, i0 + 1] * fr) * rot
            wide = wide * (8000.0 / wide.abs().amax(dim=1, keepdim=True))
            wiq = torch.stack([wide.imag, wide.real], dim=2).round().clamp(-32768, 32767).to(torch.int16).contiguous()
            self.segs = [wiq[:, i * K * 864 * CW:(i + 1) * K * 864 * CW].contiguous() for i in range(KT // K)]
            self.wiq, self.CW = wiq, CW
            self.fe = RxFrontEnd(ctx, S // C, self.lpf, max_chunks=K, carrier_freq=self.freqs, rate_factor=CW)
            self.shared = not bool(getattr(args, "per_carrier_channeliser", False))
            if self.shared:
                self.fe.set_shared_filter(True)
                self.kernel_names["k_resample"] = "k_channelise16<%d>" % C
            del wide, lo_c
        else:
            self.fe = RxFrontEnd(ctx, S, self.lpf, max_chunks=K)
        if self.group:
            self.grp = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_EQUALIZE if self.refchain else pkg.TSCLEG_DEMOD, start=(0, 0))
            for a in range(S):
                self.grp.control(a, "CMD SETTSC %d" % self.tsc)
                for tn in range(8):
                    self.grp.control(a, "CMD SETSLOT %d %d" % (tn, 5 if (tn == 0 and a % 8 == 0) else 1))
            self.slots_done = 0
        Bmax = S * (K * self.per_chunk // (156 * sps) + 2)
        self.flags = torch.zeros(Bmax, dtype=torch.uint8, device=dev)
        self.amp = torch.zeros(Bmax, 2, dtype=torch.float32, device=dev)
        self.toa = torch.zeros(Bmax, dtype=torch.float32, device=dev)
        self.soft = torch.zeros(Bmax, NSOFT, dtype=torch.float32, device=dev)
        ctx.reserve(Bmax)
        self.nbursts = 0
        self.last_nb = 0

    def step(self):
        if self.group:
            fn = (self.slots_done // 8) % (2048 * 26 * 51)
            ns, res = self.grp.pull_rxfe(self.fe, self.segs[self.seg], fn)
            self.seg = (self.seg + 1) % len(self.segs)
            self.slots_done += ns
            self.nbursts += self.S * ns
            self.last_nb = res.n_rows
            self.last_res = res
            return
        if self.fused:
            nb, _ = self.fe.push_detect_demod(self.segs[self.seg], self.tsc, self.flags, self.amp, self.toa, self.soft, detect_thresh=3.0,
                                              energy_thresh=0.0, nsoft=NSOFT, soft_stride=NSOFT)
            self.seg = (self.seg + 1) % len(self.segs)
            self.nbursts += self.S * nb
            self.last_nb = self.S * nb
            return
        if self.wide:
            self.fe.push_wideband(self.segs[self.seg])
        else:
            self.fe.push_chunk(self.segs[self.seg])
        self.seg = (self.seg + 1) % len(self.segs)
        r = self.fe.pop_raw()
        if r is None:
            return
        ps, po, pl, tn, nb = r
        B = self.S * nb
        self.ctx._chk(self.ctx.L.trxsig_detect_demod_normal_batch(
            self.ctx.h, ps, po, pl, B, self.tsc, 3.0, 0.0, self.flags.data_ptr(), self.amp.data_ptr(), self.toa.data_ptr(), None,
            self.soft.data_ptr(), None, NSOFT, NSOFT), "trxsig_detect_demod_normal_batch")
        self.nbursts += B
        self.last_nb = B

    def units_per_step(self):
        return self.S * self.K * self.per_chunk / (156.25 * self.sps)     # bursts' worth of samples per step

    def describe(self, world):
        if self.refchain:
            return {"workload": "config4 (reference chain): %d ARFCN streams/GPU x %d chunks of 864 int16 I/Q samples per step (400 kS/s), "
                                "unUSRPify + polyphaseResampleVector 65:96 with the reference's createLPF(., 961, 65) table (sps 1) + "
                                "157/156/156/156 slicing + the Transceiver group on its equalising leg: expectedCorrType per (ARFCN, slot) "
                                "(combination V on TN 0 of every 8th ARFCN), adaptive energy threshold, per-slot channel cache, "
                                "analyzeTrafficBurst + designDFE + equalizeBurst to %d soft bits" % (self.S, self.K, NSOFT),
                    "streams_per_gpu": self.S, "chunks_per_step": self.K, "bursts_per_step_per_gpu": round(self.units_per_step(), 1),
                    "sps": self.sps, "parallelism": "stream-sharded x%d, one stream set per rank (no data-path collective)" % world}
        if self.wide:
            form = ("SHARED-FILTER form: unUSRPify, sixteen partial sums of the 8001-tap Kaiser LPF's 260:768 polyphase branches on the raw "
                    "samples, sixteen complex multiply-adds per carrier (one pass for all carriers; ~1e-6 from the per-carrier form)"
                    if getattr(self, "shared", False) else
                    "per carrier unUSRPify + frequencyShift (table trig) + polyphase resample 260:768 (8001-tap Kaiser LPF)")
            return {"workload": "config4 (channeliser): %d wideband streams/GPU at 3.2 MS/s x %d carriers 400 kHz apart (= %d ARFCNs), %d chunks of "
                                "%d int16 I/Q samples per stream per step; %s behind a 1536-sample history in ONE kernel into the receive buffers, "
                                "then 157/156/156/156 slicing (pop) + TSC %d detect (thr 3.0) + demod to %d soft bits through the resampled "
                                "complex float32 stream; TSC on every slot, fixed thresholds"
                                % (self.S // self.wide, self.wide, self.S, self.K, 864 * 8, form, self.tsc, NSOFT),
                    "streams_per_gpu": self.S, "wideband_streams_per_gpu": self.S // self.wide, "carriers": self.wide,
                    "chunks_per_step": self.K, "bursts_per_step_per_gpu": round(self.units_per_step(), 1),
                    # the shared-filter form is NOT the form pinned on the reference's primitives (the per-carrier one is:
                    # --per-carrier-channeliser); the CPU baseline beside this line runs the reference's exact per-carrier chain
                    "approximate_form": bool(getattr(self, "shared", False)),
                    "approximate_form_error": ("samples within 1e-4 of the signal's scale of the per-carrier form's, soft bits within 1e-4, "
                                               "detection flags and hard bits equal (tests/test_gpu_channeliser.py::"
                                               "test_shared_filter_form_against_the_per_carrier_form)") if getattr(self, "shared", False) else None,
                    "sps": self.sps, "parallelism": "stream-sharded x%d, one stream set per rank (no data-path collective)" % world}
        return {"workload": "config4: %d ARFCN streams/GPU x %d chunks of 864 int16 I/Q samples per step (400 kS/s), unUSRPify + "
                            "polyphase resample 260:96 (961-tap Kaiser LPF) + 157/156/156/156 slicing + TSC %d detect (thr 3.0) + demod to "
                            "%d soft bits; %s" % (self.S, self.K, self.tsc, NSOFT,
                                                  "Transceiver group on the fused front end: expectedCorrType per (ARFCN, slot) -- combination V on "
                                                  "TN 0 of every 8th ARFCN (access-burst slots), combination I elsewhere --, adaptive energy threshold "
                                                  "per ARFCN replayed on the device, the resampled stream never written to HBM" if self.group else
                                                  ("one fused call, the resampled stream never written to HBM, TSC on every slot, fixed thresholds"
                                                   if self.fused else "through the resampled complex float32 stream (push + pop + detect)"))
                            + ("; soft mode TOLERANCE (flags / amp / TOA / thresholds / hard bits bit-exact, soft bits within 7.4e-5)"
                               if self.soft_mode == "tolerance" else "; soft mode EXACT"),
                "soft_mode": self.soft_mode, "streams_per_gpu": self.S, "chunks_per_step": self.K, "bursts_per_step_per_gpu": round(self.units_per_step(), 1),
                "sps": self.sps, "parallelism": "stream-sharded x%d, one stream set per rank (no data-path collective)" % world}

    def sanity(self):
        if self.group:
            r = self.grp.collect(soft=False)
            thr = r["threshold"]
            return {"softvectors_returned_frac_last_step": round(float(r["valid"].mean()), 4), "rows_last_step": int(self.last_nb),
                    "slots_last_step": int(r["valid"].shape[0]), "energy_threshold_min_max": [float(__import__("numpy").nanmin(thr)),
                                                                                            float(__import__("numpy").nanmax(thr))]}
        det = (self.flags[:self.last_nb] & self.pkg.F_DETECT) != 0
        return {"detected_frac": round(float(det.float().mean().item()), 4), "bursts_cut_last_step": int(self.last_nb)}

    def fresh_inputs(self, steps):
        return None

    def pipelined(self, steps):
        """Side measurement (never `value`): the same steps with trxsig_trxgroup_set_pipelined -- a step returns without waiting
        for its state machine, which replays on the group's side stream while the next step's detectors run."""
        if not self.group or self.refchain:
            return None
        import torch
        # (round 4: the default keeps a call on ONE stream -- the replay runs parallel in time and is short; the side-stream
        #  arrangement this mode builds on is selected for this pass only)
        self.grp.set_beside_rows(24576)
        self.grp.set_pipelined(True)
        for _ in range(max(steps // 10, 5)):
            self.step()
        self.grp.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.grp.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.grp.set_pipelined(False)
        self.grp.set_beside_rows(0)
        return {"value": round(self.units_per_step() * steps / dt / 1e6, 3), "unit": "Mbursts/s", "ms_per_step": round(dt / steps * 1e3, 4),
                "steps": steps, "what": "trxsig_trxgroup_set_beside_rows(24576) (the replay on the group's side stream) + trxsig_trxgroup_set_pipelined(1): d_valid / d_threshold of step i are complete after "
                                        "trxsig_trxgroup_sync, its replay overlaps step i+1's detectors; same values (tests/test_gpu_trxgroup.py)"}

    def stepping_state_machine(self, steps):
        """Side measurement (never `value`): the same steps with the round-4 kernels of the group's state machine (they step through every
        timeslot; trxsig_set_tuning(TRXSIG_TUNE_GROUP_REPLAY, 1)) instead of the wave-per-segment kernels that visit only the timeslots
        at which the state can move.  Same values (tests/test_gpu_trxgroup.py holds the two bit for bit against each other)."""
        if not self.group:
            return None
        import torch
        self.ctx.set_tuning(group_replay=1)
        try:
            for _ in range(max(steps // 10, 5)):
                self.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            self.ctx.set_tuning(group_replay=0)
        return {"value": round(self.units_per_step() * steps / dt / 1e6, 3), "unit": "Mbursts/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
                "what": "trxsig_set_tuning(TRXSIG_TUNE_GROUP_REPLAY, 1): k_group_replay_seg + k_group_pack + k_group_scatter / k_group_cache + k_eq_list "
                        "(round 4) instead of k_group_replay_wave / k_group_cache_wave"}

    def cpu_baseline(self, check):
        """The oracle's polyphaseResampleVector + analyzeTrafficBurst + demodulateBurst chain on ONE host core over a bounded
        sample of the streams (chunk by chunk with history, as RadioInterface::pullBuffer)."""
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oraclebind
        nchunks = min(self.KT, 125)
        if self.refchain:
            # the reference itself, equalising leg: resample chunk by chunk + ref_eq_batch (energyDetect, analyzeTrafficBurst with the
            # channel, designDFE, equalizeBurst for EVERY burst -- the per-slot channel cache of Transceiver.cpp would skip most
            # designDFE calls; the sigProcLib work per burst is otherwise the same), streams shared out over the host's cores
            import refbind
            if not refbind.available():
                return {"cpu_baseline": None}
            import subprocess
            import tempfile
            cores = host_cores()
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "sample.npz")
                np.savez(path, iq=self.iq[:, :nchunks * 864].cpu().numpy(), lpf=np.asarray(self.lpf, np.float32), sps=self.sps, tsc=self.tsc,
                         kind="config4", equalize=1)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_bench.py"), path, str(cores), "6"],
                                   capture_output=True, text=True, timeout=300)
                try:
                    return {"cpu_baseline": json.loads(r.stdout.strip().splitlines()[-1])}
                except Exception as e:
                    sys.stderr.write("reference cpu baseline unavailable: %r %s\n" % (e, r.stderr[-300:]))
                    return {"cpu_baseline": None}
        if self.wide:
            # the channeliser's work as the reference's primitives do it: per carrier frequencyShift + polyphaseResampleVector chunk by
            # chunk, slicing, analyzeTrafficBurst + demodulateBurst -- the real reference, wideband streams shared out over the cores
            import refbind
            if not refbind.available():
                return {"cpu_baseline": None}
            import subprocess
            import tempfile
            cores = host_cores()
            nch = min(nchunks, 25)                            # a bounded sample: 25 chunks of every wideband stream
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "sample.npz")
                np.savez(path, iq=self.wiq[:, :nch * 864 * self.CW].cpu().numpy(), lpf=np.asarray(self.lpf, np.float32), sps=self.sps,
                         tsc=self.tsc, kind="config4", freqs=np.asarray(self.freqs, np.float32), rate_factor=self.CW)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_bench.py"), path, str(cores), "6"],
                                   capture_output=True, text=True, timeout=600)
                try:
                    return {"cpu_baseline": json.loads(r.stdout.strip().splitlines()[-1])}
                except Exception as e:
                    sys.stderr.write("reference cpu baseline unavailable: %r %s\n" % (e, r.stderr[-300:]))
                    return {"cpu_baseline": None}
        o = oraclebind.Oracle(self.sps)
        n_bursts = 0
        t0 = time.perf_counter()
        for s in range(self.S):                               # one stream after the other until ~10 s of CPU work are done
            iq = self.iq[s].cpu().numpy()
            hist = np.zeros(192, np.complex64); rcv = []
            for c in range(nchunks):
                ch = iq[c * 864:(c + 1) * 864]
                cf = (ch[:, 1].astype(np.float32) + 1j * ch[:, 0].astype(np.float32)).astype(np.complex64)
                y = o.polyphase_resample(np.concatenate([hist, cf]), 65 * self.sps, 96, self.lpf)
                rcv.append(y[2 * 65 * self.sps:]); hist = cf[-192:]
            xs = np.concatenate(rcv)
            lens = []; pos = 0; tn = 0
            while xs.size - pos > (156 + (tn % 4 == 0)) * self.sps:
                n = (156 + (tn % 4 == 0)) * self.sps; lens.append(n); pos += n; tn = (tn + 1) % 8
            lens = np.array(lens, np.int32); off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
            r = o.normal_batch(xs, off, lens, self.tsc, nthreads=1)
            n_bursts += len(lens)
            if s == 0:
                x, ok, amp, toa, soft, lens0, pos0 = xs, r[0], r[1], r[2], r[3], lens, pos
            if time.perf_counter() - t0 > 10.0:
                break
        tt = time.perf_counter() - t0
        lens, pos = lens0, pos0
        out = {"cpu_baseline": {"value": round(n_bursts / tt / 1e6, 6), "unit": "Mbursts/s", "cores": 1, "kind": "port",
                                "sample": "%d chunks each of the first %d streams (%d bursts): polyphaseResampleVector chunk by chunk + "
                                          "analyzeTrafficBurst + demodulateBurst, oracle/sigproc_oracle.c, one thread, %.1f s"
                                          % (nchunks, s + 1, n_bursts, tt)}}
        if check:
            # the first bursts of stream 0 as the device cut them on the FIRST step are not kept; re-run one step on a fresh front end
            from openbts_ttsou_amd.frontend import RxFrontEnd
            fe = RxFrontEnd(self.ctx, self.S, self.lpf, max_chunks=self.K)
            fe.push_chunk(self.segs[0])
            xg, og, lg, tng = fe.pop_bursts()
            nb = og.numel() // self.S
            xh = xg.cpu().numpy().view(np.complex64).ravel(); o0 = int(og[0].item())
            m = min(pos, int(lg[:nb].sum().item()))
            out["oracle_check_resampled_stream0"] = bool(np.array_equal(xh[o0:o0 + m], x[:m]))
            # and the fused call's results for the same bursts against the oracle's analyzeTrafficBurst + demodulateBurst
            torch = self.torch
            fe2 = RxFrontEnd(self.ctx, self.S, self.lpf, max_chunks=self.K)
            nb2, _ = fe2.push_detect_demod(self.segs[0], self.tsc, self.flags, self.amp, self.toa, self.soft, detect_thresh=3.0,
                                           energy_thresh=0.0, nsoft=NSOFT, soft_stride=NSOFT)
            torch.cuda.synchronize()
            k = min(nb2, len(lens))
            det = ((self.flags[:k] & self.pkg.F_DETECT) != 0).cpu().numpy()
            same = np.array_equal(det, ok[:k].astype(bool)) and np.array_equal(self.toa[:k].cpu().numpy(), toa[:k]) and \
                np.array_equal(self.soft[:k].cpu().numpy()[det], soft[:k, :NSOFT][det])
            out["oracle_check_fused_stream0_first_%d" % k] = bool(same)
        # the real reference (oracle/_ref) on the box's cores: the same chain per stream, the streams shared out over processes
        import refbind
        if refbind.available():
            import subprocess
            import tempfile
            cores = host_cores()
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "sample.npz")
                np.savez(path, iq=self.iq[:, :nchunks * 864].cpu().numpy(), lpf=np.asarray(self.lpf, np.float32), sps=self.sps, tsc=self.tsc,
                         kind="config4")
                try:
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_bench.py"), path, str(cores), "6"],
                                       capture_output=True, text=True, timeout=300)
                    port = out["cpu_baseline"]
                    out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
                    out["cpu_port"] = port                  # the oracle port (one thread) beside the real reference
                except Exception as e:                      # the reference leg is optional; the port leg stands
                    sys.stderr.write("reference cpu baseline unavailable: %r\n" % (e,))
        return out


# ======================================================================================================================
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
        if have < (1 if args.rehearse_one_gpu else args.gpus):
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
    blob, _ = tdist.broadcast_tables(pkg, 4, device=None, src=0)
    lo, hi = tdist.shard_range((args.bursts or 65536) * world, rank, world)
    tmax = tdist.max_over_ranks(1.0 + rank)
    seen = tdist.ranks_seen(rank, None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": True, "value": None, "n_gpus": world, "ranks_seen": seen, "tmax": tmax,
                          "workload": args.workload,
                          "tables_fnv": int(np.frombuffer(blob.tobytes()[-8:], np.uint64)[0]) if len(blob) >= 8 else 0,
                          "shard0": [lo, hi]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 1000 (normal), 300 (rach), 100 (config4), 500 (config5)")
    ap.add_argument("--warmup", type=int, default=None, help="default: a tenth of --steps")
    ap.add_argument("--preheat-ms", type=float, default=40.0,
                    help="untimed passes before the warm-up until this much GPU time has gone by: the device needs "
                         "~15-20 ms of load to leave its idle clocks (measured: 496 Mbursts/s over the first 50 "
                         "steps of a cold run, 573 once it has ramped), whatever --warmup/--steps the caller picks")
    ap.add_argument("--bursts", type=int, default=None, help="bursts per GPU (normal, rach, config5; default 65536)")
    ap.add_argument("--streams", type=int, default=128, help="config4: ARFCN streams per GPU")
    ap.add_argument("--chunks", type=int, default=125,
                    help="config4: 864-sample chunks per stream per step.  The generated stream is a multiple of 125 chunks long "
                         "(125 chunks = 117 whole 157-156-156-156 groups) and consecutive steps push consecutive segments of it, so the "
                         "input stays aligned with the front end's burst schedule whatever K is")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="also check a sample against the CPU oracle")
    ap.add_argument("--unfused-frontend", action="store_true", help="config4: push + pop + detect through the resampled complex float32 stream instead of the fused front end")
    ap.add_argument("--wideband", type=int, default=0, help="config4: the channeliser -- streams / N wideband streams at 3.2 MS/s carrying N ARFCNs each (N <= 8)")
    ap.add_argument("--per-carrier-channeliser", action="store_true", help="config4 --wideband: the per-carrier form (bit-equal to the reference's "
                    "frequencyShift + polyphaseResampleVector per carrier) instead of the shared-filter form (one pass for all carriers, ~1e-6)")
    ap.add_argument("--reference-chain", action="store_true", help="config4: the reference's own configuration -- sps 1, its createLPF(., 961, 65) "
                    "table, the equalising Transceiver leg -- through the group on a push / pop front end")
    ap.add_argument("--stateless-frontend", action="store_true", help="config4: trxsig_rxfe_push_detect_demod_normal (TSC on every slot, fixed thresholds) instead of the Transceiver group")
    ap.add_argument("--repeats", type=int, default=None, help="extra timed repetitions of the K steps after the official one (min / median are reported beside `value`); default 4 when --steps < 100, else 0")
    ap.add_argument("--workload", choices=["normal", "rach", "config4", "config5"], default="normal",
                    help="normal = BASELINE config 2 (the headline metric); rach = config 3; config4 = resample + slice + detect "
                         "per ARFCN stream; config5 = 52M equaliser leg, fp16 storage (side measurements)")
    ap.add_argument("--chain-lag", type=int, default=None, help="A/B: path 5, tiles between detect and demodulate workgroups")
    ap.add_argument("--path", type=int, default=None, choices=[0, 1, 2, 3, 4, 5],
                    help="A/B: normal-burst implementation (trxsig_set_tuning); default = the library's")
    ap.add_argument("--spec-peak", type=int, default=0, choices=[0, 1, 2],
                    help="A/B, path 0's peak kernel: 0 = two lanes per burst (default), 1 = eight lanes, speculative, 2 = a lane per burst")
    ap.add_argument("--eq-tail", type=int, default=1, choices=[1, 2], help="A/B (config5, reference chain): 1 = k_eq_dfe4 (default), 2 = k_eq_delay + k_eq_dfe2 "
                    "(trxsig_set_tuning(TRXSIG_TUNE_EQ_TAIL))")
    ap.add_argument("--group-replay", type=int, default=0, choices=[0, 1], help="A/B (config4): the Transceiver group's state machine, 0 = the wave-per-segment "
                    "kernel (default), 1 = the kernels that step through every timeslot (trxsig_set_tuning(TRXSIG_TUNE_GROUP_REPLAY))")
    ap.add_argument("--soft-mode", choices=["tolerance", "exact"], default="tolerance",
                    help="normal / rach: demodulateBurst's arithmetic (trxsig_set_soft_mode).  tolerance (default): flags, amp, TOA and hard bits "
                         "bit-exact, soft bits within 7.4e-5 of the reference's (north_star's bar is 1e-4); exact: every soft bit IEEE-equal.  The "
                         "other mode is measured as a side field (`other_soft_mode`)")
    ap.add_argument("--no-fresh", action="store_true", help="skip the rotating-inputs side measurement")
    ap.add_argument("--no-lever", action="store_true", help="skip the workload's side measurement of an alternative arrangement (side streams, pipelined mode): "
                    "a profiler run of the default step then sees that step only")
    ap.add_argument("--generic-taps", action="store_true", help="A/B: correlators without the tap-class specialisation")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="multi-rank REHEARSAL on a one-GPU box: every rank uses cuda:0 and the collectives run over gloo on host "
                         "tensors (RCCL refuses two ranks on one device).  Exercises the whole N-rank flow -- spawn, table broadcast, "
                         "per-rank workloads, barriers, MAX over ranks, the one line -- but its number is N workloads time-sharing ONE "
                         "GPU: the line is marked \"rehearsal\" and is not a scaling measurement")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="exercise the N-rank launch path on the CPU (gloo): rendezvous, table broadcast, reductions; "
                         "no bursts are processed and no number is reported")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args, sys.argv[1:])                     # does not return
    if args.selftest_cpu:
        return selftest_cpu(args)
    if args.steps is None:
        args.steps = {"normal": 1000, "rach": 300, "config4": 100, "config5": 500}[args.workload]
    if args.warmup is None:
        args.warmup = max(1, args.steps // 10)

    import torch
    import _pkg
    pkg = _pkg.load()
    from openbts_ttsou_amd import dist as tdist

    if args.workload == "config4":
        wl = Config4(args)
    elif args.workload == "config5":
        from bench_config5 import Config5                   # (kept beside bench.py: its data generator is long)
        wl = Config5(args)
    else:
        wl = Normal(args, rach=args.workload == "rach")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        local = 0 if args.rehearse_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        rank, world = tdist.init_from_env("gloo" if args.rehearse_one_gpu else "nccl")
    else:
        rank, local = 0, 0
        torch.cuda.set_device(0)
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d -- refusing to report one under the other's label\n"
                         % (args.gpus, world))
        sys.exit(2)
    dev = torch.device("cuda", local)
    cdev = None if args.rehearse_one_gpu else dev           # where the collectives' tensors live

    # constant tables: built once on rank 0, broadcast over RCCL (xGMI), validated, then each rank's
    # context is created from the received device blob
    tuning = args.path is not None or args.chain_lag is not None or bool(args.spec_peak)   # A/B flags: libtrxsig_tune.so
    if world > 1:
        _, tbl = tdist.broadcast_tables(pkg, wl.sps, device=cdev, src=0)
        ctx = pkg.TrxSig(wl.sps, local, tables_blob=tbl.to(dev), tuning=tuning)
    else:
        ctx = pkg.TrxSig(wl.sps, local, tuning=tuning)
    ctx.use_torch_stream()
    if args.path is not None:
        ctx.set_tuning(normal_path=args.path)
    if args.chain_lag is not None:
        ctx.set_tuning(chain_lag=args.chain_lag)
    if args.generic_taps:
        ctx.set_tuning(generic_taps=1)
    if args.spec_peak:
        ctx.set_tuning(spec_peak=args.spec_peak)
    if args.eq_tail != 1:
        ctx.set_tuning(eq_tail=args.eq_tail)
    if args.group_replay:
        ctx.set_tuning(group_replay=args.group_replay)
    # demodulateBurst's arithmetic (normal, rach and config 4's demodulating legs; config 5 and the reference chain equalise instead)
    ctx.set_soft_mode(pkg.SOFT_TOLERANCE if args.soft_mode == "tolerance" else pkg.SOFT_EXACT)

    wl.setup(pkg, ctx, dev, rank, args)
    step = wl.step

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # clock ramp (untimed, not part of W): see --preheat-ms
    t_pre = time.perf_counter()
    n_pre = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        n_pre += 5
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
    elapsed_own = time.perf_counter() - t0
    elapsed = tdist.max_over_ranks(elapsed_own, cdev)
    seen = tdist.ranks_seen(rank, cdev)
    per_rank = tdist.gather_floats(elapsed_own, cdev)       # every rank's own time for the same K steps (a straggler shows here)
    # ---- more repetitions of the same K steps (outside `value`): the spread of a short run
    n_rep = args.repeats if args.repeats is not None else (4 if args.steps < 100 else 0)
    rep_ms = [elapsed / args.steps * 1e3]
    for _ in range(n_rep):
        barrier()
        tr = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        rep_ms.append(tdist.max_over_ranks(time.perf_counter() - tr, cdev) / args.steps * 1e3)
    # ---- per-kernel durations: the same steps again with every launch bracketed by HIP events
    #      (kept out of the timed region: the extra event records stretch the gaps between kernels)
    ctx.profile_enable(True)
    for _ in range(min(args.steps, 200)):
        step()
    prof = ctx.profile_collect()
    ctx.profile_enable(False)

    fresh = wl.fresh_inputs(args.steps) if (world == 1 and not args.no_fresh) else None
    prof_fresh = None
    if fresh is not None and "prof" in fresh:
        prof_fresh = fresh.pop("prof")
    piped = wl.pipelined(args.steps) if (world == 1 and hasattr(wl, "pipelined") and not args.no_lever) else None
    other_mode = wl.other_soft_mode(args.steps) if (world == 1 and hasattr(wl, "other_soft_mode") and not args.no_lever) else None
    stepping = (wl.stepping_state_machine(args.steps)
                if (world == 1 and hasattr(wl, "stepping_state_machine") and not args.no_lever and not args.group_replay) else None)
    sanity = wl.sanity()
    if rank != 0:
        return
    units = wl.units_per_step()
    value = world * units * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    # the dominant DATA kernel: a workload may name kernels that are latency chains running beside the data path (config 4's
    # k_group_replay: two waves on a side stream, a few bytes per burst) -- they are listed in kernels_ms, not priced against HBM
    beside = getattr(wl, "beside_kernels", ())
    cand = {k: v for k, v in prof.items() if k not in beside} or prof
    dom = max(cand.items(), key=lambda kv: kv[1][0]) if cand else (None, (0.0, 0))
    roof = None
    if dom[0]:
        avg_ms = dom[1][0] / max(dom[1][1], 1)
        per_unit = wl.kernel_alg.get(dom[0], wl.alg_bytes)
        launch_units = units
        if dom[0] == "k_resample" and args.workload == "config4":
            per_unit, launch_units = 4 * 864 + 8 * wl.per_chunk, wl.S * wl.K     # per stream-chunk: int16 read, c64 written
            if getattr(wl, "wide", 0):
                per_unit = 4 * 864 * 8 // wl.wide + 8 * wl.per_chunk             # the wideband chunk is read once for its carriers
        achieved = per_unit * launch_units / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": wl.kernel_names.get(dom[0], dom[0]), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None if getattr(wl, "refchain", False) else measured_traffic(wl.kernel_names.get(dom[0], dom[0]), launch_units),
                "alg_bytes_per_launch": int(per_unit * launch_units), "avg_kernel_ms": round(avg_ms, 4),
                "alg_bytes_per_unit": per_unit,
                "pipeline_achieved": round(wl.alg_bytes * units * args.steps / (ev_ms * 1e-3) / 1e9, 1),
                "pipeline_frac": round(wl.alg_bytes * units * args.steps / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "kernels_ms": {wl.kernel_names.get(k, k): round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}}
        alg_read = getattr(wl, "alg_read", None)
        if alg_read:
            # the bar of SURVEY 8d is on the STEP's read roofline: algorithmic bytes read per unit x units / step time / 8 TB/s
            roof["read_frac"] = round(alg_read * units * args.steps / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # HBM bytes the whole step moved, from the committed counter passes (profiles/traffic.json), beside the algorithmic bytes
        tr = {wl.kernel_names.get(k, k): measured_traffic(wl.kernel_names.get(k, k), units) for k in prof if k not in beside}
        if tr and all(v is not None for v in tr.values()) and not getattr(wl, "refchain", False):
            roof["pipeline_traffic"] = {"bytes_per_step": int(sum(tr.values())), "per_kernel": tr,
                                        "over_algorithmic": round(sum(tr.values()) / float(wl.alg_bytes * units), 3)}
        if any(k in prof for k in beside):
            roof["beside_the_data_path"] = {wl.kernel_names.get(k, k): round(prof[k][0] / max(prof[k][1], 1), 4) for k in beside if k in prof}
    roof_fresh = None
    if prof_fresh and dom[0] in prof_fresh:
        # the same kernel on inputs it has not seen a step ago (three batches in rotation, 1 GB > the 256 MB memory-side cache)
        f_ms = prof_fresh[dom[0]][0] / max(prof_fresh[dom[0]][1], 1)
        per_unit = wl.kernel_alg.get(dom[0], wl.alg_bytes)
        ach = per_unit * units / (f_ms * 1e-3) / 1e9
        roof_fresh = {"bound": "hbm", "kernel": wl.kernel_names.get(dom[0], dom[0]), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "avg_kernel_ms": round(f_ms, 4),
                      "alg_bytes_per_launch": int(per_unit * units), "inputs_in_rotation": 3,
                      "pipeline_frac": round(wl.alg_bytes * fresh["value"] * 1e6 / 1e9 / HBM_PEAK_GBS, 4),
                      "read_frac": (round(wl.alg_read * fresh["value"] * 1e6 / 1e9 / HBM_PEAK_GBS, 4) if getattr(wl, "alg_read", None) else None),
                      "kernels_ms": {wl.kernel_names.get(k, k): round(v[0] / max(v[1], 1), 4) for k, v in prof_fresh.items()}}
    out = {
        "metric": "Mbursts/s (156.25-sym @ 4 sps) demod+detect", "value": round(value, 3), "unit": "Mbursts/s",
        "n_gpus": world, "ranks_seen": seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
        "config": wl.describe(world),
        "hip_event_ms_per_step": round(ev_ms / args.steps, 4), "preheat_steps": n_pre,
        "roofline": roof, "fresh_inputs": fresh, "roofline_fresh": roof_fresh,
        "per_rank": [{"rank": i, "value": round(units * args.steps / t / 1e6, 3), "ms_per_step": round(t / args.steps * 1e3, 4)}
                     for i, t in enumerate(per_rank)],
        "repetitions": {"n": len(rep_ms), "ms_per_step": [round(v, 4) for v in rep_ms], "min_ms_per_step": round(min(rep_ms), 4),
                        "median_ms_per_step": round(sorted(rep_ms)[len(rep_ms) // 2], 4),
                        "median_value": round(world * units / (sorted(rep_ms)[len(rep_ms) // 2] * 1e-3) / 1e6, 3),
                        "note": "`value` is the first repetition (the contract's K steps); the others follow it back to back"},
    }
    out.update(sanity)
    if piped:
        out[getattr(wl, "pipelined_key", "pipelined")] = piped
    if other_mode:
        out["other_soft_mode"] = other_mode
    if stepping:
        out["stepping_state_machine"] = stepping
    if args.rehearse_one_gpu:
        out["rehearsal"] = "all %d ranks shared cuda:0 (gloo collectives): launch-path check, not a scaling number" % world
    if not args.no_cpu_baseline and world == 1:
        out.update(wl.cpu_baseline(args.check))
        if isinstance(out.get("cpu_baseline"), dict):
            out["cpu_baseline"].update(host_info())         # `cores` = workers used; these say what the box is
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
