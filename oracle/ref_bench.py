"""Time the REAL reference (oracle/_ref/libref_sigproc.so = Transceiver/sigProcLib.cpp compiled in place by oracle/Makefile)
on a sample of the bench workload: analyzeTrafficBurst + demodulateBurst per burst, P worker PROCESSES (the reference
keeps its tables in process globals), each looping over its own contiguous slice of the sample.  Test infrastructure:
called by bench.py's cpu_baseline leg as a child process (it never touches the GPU).

    python oracle/ref_bench.py sample.npz P seconds      -> one JSON line
sample.npz: x complex64 (packed bursts), off int32, length int32, sps, tsc."""
import json
import multiprocessing as mp
import sys
import time

import numpy as np


def worker(path, lo, hi, reps, start, done):
    import refbind
    d = np.load(path)
    r = refbind.Ref(int(d["sps"]))
    x, off, length, tsc = d["x"], d["off"][lo:hi], d["length"][lo:hi], int(d["tsc"])
    base = int(off[0])
    xs = np.ascontiguousarray(x[base:int(off[-1] + length[-1])])
    off = (off - base).astype(np.int32)
    r.normal_batch(xs, off, length, tsc)                 # warm (page in, allocator)
    start.wait()
    for _ in range(reps):
        r.normal_batch(xs, off, length, tsc)
    done.wait()


def main():
    path, P, seconds = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    import refbind
    d = np.load(path)
    B = len(d["off"])
    # single-process calibration on the first 512 bursts
    r = refbind.Ref(int(d["sps"]))
    n1 = min(512, B)
    off1, len1 = d["off"][:n1], d["length"][:n1]
    x1 = np.ascontiguousarray(d["x"][:int(off1[-1] + len1[-1])])     # (an .npz member is re-read on every access)
    r.normal_batch(x1, off1, len1, int(d["tsc"]))
    t0 = time.perf_counter()
    r.normal_batch(x1, off1, len1, int(d["tsc"]))
    per_burst = (time.perf_counter() - t0) / n1
    per_pass = per_burst * (B / P)
    reps = max(1, int(seconds / max(per_pass, 1e-4)))
    ctx = mp.get_context("fork")
    start, done = ctx.Barrier(P + 1), ctx.Barrier(P + 1)
    edges = [B * i // P for i in range(P + 1)]
    procs = [ctx.Process(target=worker, args=(path, edges[i], edges[i + 1], reps, start, done)) for i in range(P)]
    for p in procs:
        p.start()
    start.wait(timeout=120)
    t0 = time.perf_counter()
    done.wait(timeout=600)
    tt = time.perf_counter() - t0
    for p in procs:
        p.join()
    print(json.dumps({"value": round(B * reps / tt / 1e6, 6), "unit": "Mbursts/s", "cores": P, "kind": "reference",
                      "single_thread_Mbursts_per_s": round(1e-6 / per_burst, 6),
                      "sample": "%d passes over the first %d bursts of the GPU batch (analyzeTrafficBurst + demodulateBurst of "
                                "Transceiver/sigProcLib.cpp compiled in place, oracle/_ref/libref_sigproc.so, %d processes, %.1f s)"
                                % (reps, B, P, tt)}))


if __name__ == "__main__":
    main()
