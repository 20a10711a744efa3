"""Config 2 with the compute units PARTITIONED between the detectors and the demodulator.

TRXSIG_TUNE_DEMOD_BESIDE runs the demodulator of call i beside the correlator of call i+1; with TRXSIG_TUNE_BESIDE_DET_CUS = x the
two streams are created with hipExtStreamCreateWithCUMask: the detectors (k_tsc_corr, k_tsc_peak2) get x CUs, k_demod the other
256 - x.  For every x (and both readings of the mask's bit order) this prints
  * the kernels ALONE on their share of the machine (a synchronise after every call: nothing co-runs) -- where k_demod stops
    being HBM-bound as CUs are taken away, what the VALU-bound correlator costs on x CUs;
  * the pipelined step (K calls back to back, one synchronise at the end) and whether the outputs equal the default path's.
Run on the GPU box:  python tools/cu_split.py [--bursts 65536] [--steps 300]
One JSON line per configuration on stdout (-> profiles/r04_cu_split.txt)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bursts", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--splits", type=str, default="0,64,96,128,160,192")
    ap.add_argument("--layouts", type=str, default="0,1")
    ap.add_argument("--nodeps", type=int, default=0, help="TIMING EXPERIMENT (racy results, tuning library): 1 = drop the 'inputs ready' "
                    "dependency, 2 = drop every cross-stream dependency -- the cost of the dependencies themselves")
    args = ap.parse_args()
    import torch
    import _pkg
    pkg = _pkg.load()
    from openbts_ttsou_amd import synth
    dev = torch.device("cuda:0")
    sps, tsc, B, NS = 4, 2, args.bursts, 148
    ctx = pkg.TrxSig(sps, 0, tuning=bool(args.nodeps))
    ctx.use_torch_stream()
    if args.nodeps:
        ctx._chk(ctx.L.trxsig_set_tuning(ctx.h, 10, args.nodeps), "nodeps")
    x, off, length, meta = synth.normal_batch_torch(sps, B, tsc, seed=0xB5E55ED0, device=dev)
    xf = torch.view_as_real(x).contiguous()
    flags = torch.zeros(B, dtype=torch.uint8, device=dev); amp = torch.zeros(B, 2, device=dev); toa = torch.zeros(B, device=dev)
    soft = torch.zeros(B, NS, device=dev)
    ctx.reserve(B)

    def step():
        ctx.detect_demod_normal(xf, off, length, tsc, flags, amp, toa, soft, detect_thresh=3.0, energy_thresh=0.0, nsoft=NS, soft_stride=NS)

    def sync():
        ctx.synchronize(); torch.cuda.synchronize()

    def timed(k):
        sync()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        sync()
        return (time.perf_counter() - t0) / k * 1e3

    def alone(k):
        """per-kernel averages with a synchronise after every call: each kernel has its share of the machine to itself"""
        ctx.profile_enable(True)
        for _ in range(k):
            step(); sync()
        pf = ctx.profile_collect()
        ctx.profile_enable(False)
        return {n: round(v[0] / max(v[1], 1) * 1e3, 2) for n, v in pf.items()}

    # clock ramp + the default path
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        for _ in range(10):
            step()
        sync()
    base_ms = timed(args.steps)
    ref_soft, ref_flags, ref_toa = soft.clone(), flags.clone(), toa.clone()
    print(json.dumps({"config": "default (three launches, one stream)", "ms_per_step": round(base_ms, 4),
                      "Mbursts_per_s": round(B / base_ms / 1e3, 1), "kernels_us_alone": alone(50)}), flush=True)
    for layout in [int(v) for v in args.layouts.split(",")]:
        for x_cus in [int(v) for v in args.splits.split(",")]:
            if x_cus == 0 and layout != 0:
                continue
            ctx.set_tuning(demod_beside=1, beside_det_cus=x_cus, cu_layout=layout)
            for _ in range(20):
                step()
            sync()
            k_alone = alone(50)
            for _ in range(30):
                step()
            ms = [timed(args.steps) for _ in range(3)]
            same = bool(torch.equal(soft, ref_soft) and torch.equal(flags, ref_flags) and torch.equal(toa, ref_toa))
            print(json.dumps({"config": "demod beside the next call's detectors", "det_cus": x_cus, "demod_cus": (256 - x_cus) if x_cus else 256,
                              "cu_layout": layout, "masks": bool(x_cus), "nodeps": args.nodeps, "kernels_us_alone": k_alone,
                              "ms_per_step": [round(v, 4) for v in ms], "Mbursts_per_s": round(B / min(ms) / 1e3, 1),
                              "same_outputs_as_default": same}), flush=True)
            ctx.set_tuning(demod_beside=0)
    ctx.set_tuning(beside_det_cus=0, cu_layout=0)
    for prio in (1, 2):                                      # no masks: the demodulator's stream above / below the detectors'
        ctx.set_tuning(demod_beside=1, beside_priority=prio)
        for _ in range(30):
            step()
        ms = [timed(args.steps) for _ in range(3)]
        same = bool(torch.equal(soft, ref_soft) and torch.equal(flags, ref_flags) and torch.equal(toa, ref_toa))
        print(json.dumps({"config": "demod beside the next call's detectors, no masks", "demod_stream_priority": {1: "highest", 2: "lowest"}[prio],
                          "nodeps": args.nodeps, "ms_per_step": [round(v, 4) for v in ms], "Mbursts_per_s": round(B / min(ms) / 1e3, 1),
                          "same_outputs_as_default": same}), flush=True)
        ctx.set_tuning(demod_beside=0)
    ctx.set_tuning(beside_priority=0)
    print(json.dumps({"config": "default again", "ms_per_step": round(timed(args.steps), 4)}), flush=True)


if __name__ == "__main__":
    main()
