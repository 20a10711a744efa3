"""profiles/traffic.json from the committed counter summaries profiles/r03_pmc_<workload>.txt (r02_ where round 3 did not
profile the workload; tools/pmc.sh via tools/r04_profiles.sh): HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE KiB (FETCH_SIZE doubled per the gfx950 correction
of MI355X_MICROARCH.md).  bench.py reads the file for `roofline.traffic`.
    python tools/traffic_from_pmc.py"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
# (file, kernel-name prefix in the summary, key in traffic.json, units per launch, extra fields)
WANT = [("normal", "k_demod<4", "k_demod", 65536, {}), ("normal", "k_tsc_corr<4", "k_tsc_corr", 65536, {}),
        ("normal", "k_tsc_peak2<4", "k_tsc_peak2", 65536, {}), ("normal", "k_tsc_peak2<4", "k_tsc_peak", 65536, {}),
        ("rach", "k_rach_front<4", "k_rach_front", 65536, {}), ("rach", "k_rach_peak2<4", "k_rach_peak2", 65536, {}),
        ("config5", "k_eq_detect52<", "k_eq_detect", 65536, {"kernel": "k_eq_detect52"}),
        ("config5", "k_eq_dfe4<", "k_eq_dfe4", 65536, {}),
        ("config4", "k_demod_rx<4", "k_demod_rx", 59904, {}), ("config4", "k_tsc_corr_rx<4", "k_tsc_corr_rx", 59904, {}),
        ("config4", "k_rach_front_rx<4", "k_rach_front_rx", 59904, {"note": "per step of the group bench (the access-burst rows only: ~500 bursts)"}),
        ("config4", "k_group_replay_seg", "k_group_replay", 59904, {"kernel": "k_group_replay_seg<16>", "note": "per step: 128 ARFCNs x 468 slots"}),
        ("config5", "k_eq_dfe4<", "k_eq_dfe", 65536, {"kernel": "k_eq_dfe4"}),
        ("config4_unfused", "k_rx_resample", "k_rx_resample", 128 * 125, {"note": "units = stream-chunks (128 streams x 125 chunks per launch)"}),
        ("config4_unfused", "k_rx_resample", "k_resample", 128 * 125, {"kernel": "k_rx_resample", "note": "units = stream-chunks (128 streams x 125 chunks per launch)"})]


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.strip():
            continue
        if not line.startswith(" "):
            cur = line.strip(); out.setdefault(cur, {})
        else:
            m = re.match(r"\s+(\w+)\s+avg\s+([0-9.]+)", line)
            if m and cur:
                out[cur][m.group(1)] = float(m.group(2))
    return out


def main():
    kernels = {}
    cache = {}
    for wl, prefix, key, units, extra in WANT:
        if wl not in cache:
            newest = [os.path.join(P, "r%02d_pmc_%s.txt" % (r, wl)) for r in (5, 4, 3, 2)]   # the latest round that profiled this workload
            cache[wl] = parse([f for f in newest if os.path.exists(f)][0])
        hit = [k for k in cache[wl] if k.startswith(prefix) and "FETCH_SIZE" in cache[wl][k] and "WRITE_SIZE" in cache[wl][k]]
        if not hit:
            raise SystemExit("no counters for %s in the pmc summary of %s" % (prefix, wl))
        c = cache[wl][hit[0]]
        e = dict(extra)
        e.update(fetch_kb=round(c["FETCH_SIZE"], 1), write_kb=round(c["WRITE_SIZE"], 1),
                 hbm_bytes_per_launch=int(round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)))
        e["bursts_per_launch"] = units
        kernels[key] = e
    doc = {"source": "profiles/r05_pmc_<workload>.txt (r04_, r03_, r02_ for workloads a later round did not profile; `normal` = the tolerance-mode demodulator, the headline's) (rocprofv3 --pmc FETCH_SIZE / "
                     "WRITE_SIZE in separate passes with --kernel-trace only, tools/pmc.sh via tools/r05_profiles.sh; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md; units KiB); "
                     "rebuilt by tools/traffic_from_pmc.py", "kernels": kernels}
    json.dump(doc, open(os.path.join(P, "traffic.json"), "w"), indent=1)
    for k, v in kernels.items():
        print("%-14s %6.1f MB per launch" % (k, v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
