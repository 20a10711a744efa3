"""Config 4's default call (Transceiver group on the fused front end, 128 ARFCNs x 125 chunks per step) under the round-5 switches:
soft mode (exact / tolerance) and the state machine on the side stream (set_beside_rows).  (Round 5 also measured the access-burst
class on the side stream beside the normal-burst detectors, with and without a high-priority stream: 197-203 against 192-194 us per
step, profiles/r05_config4_ab_rach_beside.txt -- not kept.)  ms per step, Mbursts/s, per-kernel HIP-event averages; one JSON line per setting.
  python tools/config4_ab.py [--streams 128 --chunks 125 --steps 100]          (-> profiles/r05_config4_ab.txt)"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=128)
    ap.add_argument("--chunks", type=int, default=125)
    ap.add_argument("--steps", type=int, default=100)
    a = ap.parse_args()
    import torch
    import _pkg
    pkg = _pkg.load()
    import bench
    dev = torch.device("cuda:0")
    for soft in ("exact", "tolerance"):
        args = argparse.Namespace(streams=a.streams, chunks=a.chunks, soft_mode=soft, bursts=None)
        wl = bench.Config4(args)
        ctx = pkg.TrxSig(wl.sps, 0); ctx.use_torch_stream()
        ctx.set_soft_mode(pkg.SOFT_TOLERANCE if soft == "tolerance" else pkg.SOFT_EXACT)
        wl.setup(pkg, ctx, dev, 0, args)
        for rach_beside in (0,):
            for rows in (0, 24576):
                wl.grp.set_beside_rows(rows)
                t_end = time.perf_counter() + 0.2
                while time.perf_counter() < t_end:
                    for _ in range(5):
                        wl.step()
                    wl.grp.sync(); torch.cuda.synchronize()
                times = []
                for _ in range(3):
                    wl.grp.sync(); torch.cuda.synchronize(); n0 = wl.nbursts
                    t0 = time.perf_counter()
                    for _ in range(a.steps):
                        wl.step()
                    wl.grp.sync(); torch.cuda.synchronize()
                    times.append(((time.perf_counter() - t0) / a.steps * 1e3, (wl.nbursts - n0) / a.steps))
                ms, per = sorted(times)[1]
                ctx.profile_enable(True)
                for _ in range(40):
                    wl.step()
                wl.grp.sync(); torch.cuda.synchronize()
                pf = ctx.profile_collect(); ctx.profile_enable(False)
                print(json.dumps({"soft_mode": soft, "rach_beside": rach_beside, "replay_beside_rows": rows, "ms_per_step": round(ms, 4),
                                  "Mbursts_per_s": round(per / ms / 1e3, 1),
                                  "kernels_us": {n: round(v[0] / max(v[1], 1) * 1e3, 1) for n, v in pf.items() if v[1]}}), flush=True)
        wl.grp.close(); wl.fe.close(); ctx.close()


if __name__ == "__main__":
    main()
