"""Soak of the Transceiver group (trxsig_trxgroup_*) against S independent single-burst objects (trxsig_trx_pull_radio_vector):
random schedules are those of tests/test_gpu_trxgroup.py (combinations I / II / IV / V / VI / VII / NONE, four TSCs), here over
several seeds, start frames (incl. across the hyperframe wrap), random call sizes, both TSC legs and the pipelined mode --
every SoftVector, RSSI, timing offset, verdict and the threshold after every burst compared for equality.
    python tools/group_soak.py [seeds [first seed]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import _pkg
import test_gpu_trxgroup as T
import transceiver_model as tm

pkg = _pkg.load()
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # first seed (a second run with other seeds: tools/group_soak.py 8 8)
total = 0
t00 = time.time()
for seed in range(seed0, seed0 + nseeds):
    rng = np.random.default_rng(1000 + seed)
    for sps, leg in ((4, 1), (1, 0)):
        S = 128
        frames = 64 if sps == 4 else 80
        n_slots = 8 * frames
        fn0 = int(rng.choice([17, 51 * 26 * 7 + 3, tm.HYPERFRAME - 25, int(rng.integers(0, tm.HYPERFRAME))]))
        tn0 = int(rng.integers(0, 8))
        q0 = int(rng.integers(50, 200))
        x, ctype = T.build_cells(sps, S, n_slots, fn0, tn0, seed=5000 + 17 * seed + sps, quiet_slots=(q0, q0 + 440))
        calls = tuple(int(v) for v in rng.choice([1, 2, 5, 8, 16, 31, 64, 200, 450, 512], size=8))
        piped = bool(leg == 1 and seed % 2 == 1)
        out, responses, final_thr = T.run_group(pkg, sps, leg, S, n_slots, fn0, tn0, x, calls, pipelined=piped)
        objs = [pkg.TrxHost(sps, 0, start=(fn0, tn0), tsc_leg=leg) for _ in range(S)]
        for a in range(S):
            assert T.configure(objs[a].control, a) == responses[a]
        seen = T.check_against("single", lambda a, b, tn, fn: objs[a].pull_radio_vector(b, tn, fn), ctype, x, sps, out, range(S), fn0, tn0,
                               lambda a: objs[a].energy_threshold)
        assert np.array_equal(final_thr, np.array([o.energy_threshold for o in objs]))
        for o in objs:
            o.close()
        n = sum(seen.values())
        total += n
        print("seed %d sps %d leg %s%s fn0 %d tn0 %d calls %s: %d bursts identical (%s)  [%.0f s]" % (
            seed, sps, "demod" if leg else "equalize", " pipelined" if piped else "", fn0, tn0, calls, n, seen, time.time() - t00), flush=True)
print("group soak: %d bursts on correlating slots, every output identical to single-burst objects" % total)
