"""bench.py --workload config5: BASELINE config 5, the Transceiver52M receive leg at one sample per symbol with the
bursts stored as fp16 I/Q in HBM (values are fp16-exact integers, |v| <= 2048, so the CPU oracle sees the same
numbers): energy gate (stride 4) -> windowed midamble correlation (maxTOA 4) with channel estimate -> designDFE(Nf = 7)
-> delayVector + equalizeBurst, one call of trxsig_equalize_normal_batch_fmt per step.  Every other burst went through a
{1, 0.4+0.2j} two-path channel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


class Config5:
    sps, tsc = 1, 6
    dtype = "fp16 storage, f32 arithmetic"

    def __init__(self, args):
        self.B = args.bursts or 65536
        self.alg_bytes = 4 * 625 // 4 + 4 * 156 + 16           # 4*156.25 B of fp16 I/Q read; 156 soft bits + flag / amp / TOA written
        # per-kernel algorithmic bytes per burst: k_eq_detect52 reads the 26-sample window + 20 energy samples (fp16) and writes
        # flag / amp / TOA / toa_eq + 12 taps; k_eq_dfe4 (scaleVector + delayVector + equalizeBurst in one kernel) reads the burst
        # (fp16), amp / TOA / flag and the taps and writes 156 soft bits.  (--eq-tail 2 = trxsig_set_tuning(TRXSIG_TUNE_EQ_TAIL, 2), the two-kernel route: k_eq_delay reads the
        # burst and writes 157 delayed c64, k_eq_dfe2 reads those + the taps and writes the soft bits.)
        two = getattr(args, "eq_tail", 1) == 2
        self.kernel_alg = {"k_eq_detect": 4 * 46 + 17 + 4 + 96, "k_eq_delay": 625 + 13 + 8 * 157,
                           "k_eq_dfe": (8 * 157 + 96 + 4 * 156) if two else (625 + 13 + 96 + 4 * 156)}
        self.kernel_names = {"k_eq_dfe": "k_eq_dfe2" if two else "k_eq_dfe4", "k_eq_detect": "k_eq_detect52"}

    def setup(self, pkg, ctx, dev, rank, args):
        import torch
        from openbts_ttsou_amd import synth
        self.pkg, self.ctx, self.dev, self.torch = pkg, ctx, dev, torch
        B = self.B
        x, off, length, meta = synth.normal_batch_torch(1, B, self.tsc, seed=5 + rank, device=dev, sigmas=(0.02, 0.1), max_delay=1.0)
        xe = x.clone()
        odd = torch.arange(1, B, 2, device=dev)
        st = off[odd].long(); ln = length[odd].long()
        for k in range(1, 157):
            sel = k < ln
            xe[st[sel] + k] = x[st[sel] + k] + (0.4 + 0.2j) * x[st[sel] + k - 1]
        xr = torch.view_as_real(xe)
        scale = 2000.0 / float(xr.abs().max().item())
        q = torch.clamp(torch.round(xr * scale), -2048, 2048)
        self.half = q.to(torch.float16).contiguous()            # [n, 2] half: what the kernels read
        self.xq = q.contiguous()                                # the same values in float32 (CPU baseline / checks)
        self.off, self.length, self.meta = off, length, meta
        self.flags = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.amp = torch.zeros(B, 2, device=dev); self.toa = torch.zeros(B, device=dev)
        self.w = torch.zeros(B, 7, 2, device=dev); self.b = torch.zeros(B, 5, 2, device=dev)
        self.soft = torch.zeros(B, 157, device=dev)
        ctx.reserve(B)

    def step(self):
        self.ctx.equalize_normal(self.half, self.off, self.length, self.tsc, self.flags, self.amp, self.toa, self.soft, w=self.w, b=self.b,
                                 energy_thresh=10.0, variant52m=True, max_toa=4, nsoft=156, soft_stride=157, fp16=True)

    def units_per_step(self):
        return self.B

    def describe(self, world):
        return {"workload": "config5: %d bursts/GPU at 1 sample/symbol stored as fp16 I/Q, 52M leg: energy gate + windowed TSC %d "
                            "correlation (maxTOA 4) + channel estimate + designDFE(Nf 7) + equalizeBurst to 156 soft bits" % (self.B, self.tsc),
                "bursts_per_gpu": self.B, "sps": 1, "parallelism": "burst-sharded x%d (no data-path collective)" % world}

    def sanity(self):
        torch = self.torch
        det = (self.flags & self.pkg.F_DETECT) != 0
        hard = (self.soft[:, :148] > 0.5).to(torch.uint8)
        ber = float((hard[det] != self.meta["bits"][det]).float().mean().item())
        return {"detected_frac": round(float(det.float().mean().item()), 4), "bit_error_rate": round(ber, 6)}

    def fresh_inputs(self, steps):
        return None

    def cpu_baseline(self, check):
        """The 52M oracle on ONE host core over the first bursts: energyDetect, analyzeTrafficBurst(requestChannel, maxTOA 4),
        scaleVector, designDFE, equalizeBurst; with --check every one of them value-exact against the device results."""
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oraclebind
        o = oraclebind.Oracle(1, variant52m=True)
        n = min(self.B, 32768)
        off = self.off[:n].cpu().numpy(); length = self.length[:n].cpu().numpy()
        xh = self.xq[:int(off[-1] + length[-1])].cpu().numpy().view(np.complex64).ravel()
        fl = self.flags[:n].cpu().numpy(); soft_d = self.soft[:n].cpu().numpy()
        thr = 10.0
        same = True
        t0 = time.perf_counter()
        for i in range(n):
            s = xh[off[i]:off[i] + length[i]]
            ok_e, _ = o.energy_detect(s, 20, thr)
            soft = None
            a = None
            if ok_e:
                a = o.analyze_traffic(s, self.tsc, 3.0, req_chan=True, max_toa=4)
                if a["ok"]:
                    am = a["amp"]
                    n2 = np.float32(np.float32(am.imag * am.imag) + np.float32(am.real * am.real))
                    inv = complex(np.float32(am.real / n2), np.float32(-am.imag / n2))
                    snr = np.float32(np.float64(n2) / (np.float64(np.float32(thr * thr)) + 1.0))
                    w, b = o.design_dfe(o.scale_vector(a["chan"], inv), float(snr), 7)
                    soft = o.equalize(o.scale_vector(s, inv), np.float32(a["toa"] - a["chan_off"]), w, b)
            if check:
                det = bool(a and a["ok"])
                same = same and (bool(fl[i] & self.pkg.F_ENERGY) == ok_e) and (bool(fl[i] & self.pkg.F_DETECT) == det)
                if det:
                    same = same and np.array_equal(soft_d[i, :156], soft[:156])
        tt = time.perf_counter() - t0
        out = {"cpu_baseline": {"value": round(n / tt / 1e6, 6), "unit": "Mbursts/s", "cores": 1, "kind": "port",
                                "sample": "the first %d bursts of the GPU batch, one by one through the 52M oracle (energyDetect + "
                                          "analyzeTrafficBurst(requestChannel) + designDFE + equalizeBurst, oracle/sigproc_oracle.c behind "
                                          "ctypes), one thread, %.1f s" % (n, tt)}}
        if check:
            out["oracle_check_first_%d" % n] = bool(same)
        # the real reference (Transceiver52M/sigProcLib.cpp compiled in place) on the box's cores, the calls strung together as
        # Transceiver::pullRadioVector does (oracle/ref_driver.cpp ref_eq_batch), one process per core
        import refbind
        if refbind.available("52m"):
            import json
            import subprocess
            import tempfile
            from bench import host_cores
            cores = host_cores()
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "sample.npz")
                np.savez(path, x=xh, off=off, length=length, sps=1, tsc=self.tsc, kind="config5", energy_thresh=thr, max_toa=4)
                try:
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_bench.py"), path, str(cores), "6"],
                                       capture_output=True, text=True, timeout=300)
                    port = out["cpu_baseline"]
                    out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
                    out["cpu_port"] = port
                except Exception as e:
                    sys.stderr.write("reference cpu baseline unavailable: %r\n" % (e,))
        return out
