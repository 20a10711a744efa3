"""Time the REAL reference (oracle/_ref/libref_sigproc.so = Transceiver/sigProcLib.cpp compiled in place by oracle/Makefile)
on a sample of the bench workload: analyzeTrafficBurst + demodulateBurst per burst, P worker PROCESSES (the reference
keeps its tables in process globals), each looping over its own contiguous slice of the sample.  Test infrastructure:
called by bench.py's cpu_baseline leg as a child process (it never touches the GPU).

    python oracle/ref_bench.py sample.npz P seconds      -> one JSON line
sample.npz: x complex64 (packed bursts), off int32, length int32, sps, tsc; kind (optional): "normal" (default),
"rach" (detectRACHBurst + demodulateBurst), "config5" (the Transceiver52M equalised leg: energyDetect + analyzeTrafficBurst with
the channel response + designDFE + equalizeBurst, oracle/_ref/libref_sigproc52m.so; energy_thresh, max_toa in the file), or "config4": iq int16 [S][K*864][2] (Q first), lpf float32 taps -- per stream
unUSRPify + polyphaseResampleVector chunk by chunk behind a 192-sample history (RadioInterface::pullBuffer) + the
157-156-156-156 slicing + analyzeTrafficBurst + demodulateBurst, the streams shared out over the processes.  With
`freqs` (float32 [C], radians per wideband sample) and `rate_factor` in a config4 file, iq holds WIDEBAND streams
[Sw][K*864*rate_factor][2] and every carrier of a stream is mixed down with frequencyShift (chunk by chunk, the running
phase handed on) and resampled 65*sps : 96*rate_factor behind a 192*rate_factor-sample history before the same slicing +
detection: the channeliser's work as the reference's primitives do it."""
import json
import multiprocessing as mp
import sys
import time

import numpy as np


def kind_of(d):
    return str(d["kind"]) if "kind" in d.files else "normal"


def config4_stream(r, iq, lpf, sps, tsc, equalize=False, freqs=None, rate_factor=1):
    """One stream through the reference: returns the number of bursts it cut and detected + demodulated (equalize: the
    equalised leg, ref_eq_batch, instead of analyzeTrafficBurst + demodulateBurst).  freqs: a wideband stream, every carrier
    in turn (frequencyShift + polyphaseResampleVector per chunk)."""
    if freqs is not None:
        total = 0
        for f in freqs:
            total += config4_carrier(r, iq, lpf, sps, tsc, float(f), rate_factor)
        return total
    nchunks = iq.shape[0] // 864
    hist = np.zeros(192, np.complex64)
    rcv = []
    for c in range(nchunks):
        ch = iq[c * 864:(c + 1) * 864]
        cf = (ch[:, 1].astype(np.float32) + 1j * ch[:, 0].astype(np.float32)).astype(np.complex64)   # unUSRPifyVector: I/Q swapped
        y = r.polyphase_resample(np.concatenate([hist, cf]), 65 * sps, 96, lpf)
        rcv.append(y[2 * 65 * sps:]); hist = cf[-192:]
    return cut_and_detect(r, np.concatenate(rcv), sps, tsc, equalize)


def config4_carrier(r, iq, lpf, sps, tsc, freq, CW):
    n = 864 * CW
    nchunks = iq.shape[0] // n
    hist = np.zeros(192 * CW, np.complex64)
    rcv = []
    phase = np.float32(0.0)
    for c in range(nchunks):
        ch = iq[c * n:(c + 1) * n]
        cf = (ch[:, 1].astype(np.float32) + 1j * ch[:, 0].astype(np.float32)).astype(np.complex64)
        # frequencyShift in blocks of 256 samples, the phase wrapped in between: over a whole chunk the running phase reaches
        # ~10^4 rad and expjLookup's subtract-one range reduction, not the mixing, would be what is timed
        for b0 in range(0, n, 256):
            cf[b0:b0 + 256], phase = r.frequency_shift(cf[b0:b0 + 256], freq, phase)
            phase = np.float32(np.fmod(float(phase), 2.0 * np.pi))
        y = r.polyphase_resample(np.concatenate([hist, cf]), 65 * sps, 96 * CW, lpf)
        rcv.append(y[2 * 65 * sps:]); hist = cf[-192 * CW:]
    return cut_and_detect(r, np.concatenate(rcv), sps, tsc, False)


def cut_and_detect(r, xs, sps, tsc, equalize):
    lens = []; pos = 0; tn = 0
    while xs.size - pos > (156 + (tn % 4 == 0)) * sps:
        n = (156 + (tn % 4 == 0)) * sps; lens.append(n); pos += n; tn = (tn + 1) % 8
    lens = np.array(lens, np.int32); off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    if equalize:
        r.eq_batch(xs, off, lens, tsc, 3.0, 0.0, 4)
    else:
        r.normal_batch(xs, off, lens, tsc)
    return len(lens)


def worker(path, lo, hi, reps, start, done):
    import refbind
    d = np.load(path)
    kind = kind_of(d)
    r = refbind.Ref(int(d["sps"]), variant="52m" if kind == "config5" else "")
    if kind == "config4":
        iq, lpf, sps, tsc = np.ascontiguousarray(d["iq"][lo:hi]), d["lpf"], int(d["sps"]), int(d["tsc"])
        eq = "equalize" in d.files and int(d["equalize"]) != 0
        freqs, CW = (d["freqs"], int(d["rate_factor"])) if "freqs" in d.files else (None, 1)
        config4_stream(r, iq[0][:864 * 4 * CW], lpf, sps, tsc, eq, freqs, CW)  # warm
        start.wait()
        for _ in range(reps):
            for s in range(hi - lo):
                config4_stream(r, iq[s], lpf, sps, tsc, eq, freqs, CW)
        done.wait()
        return
    x, off, length, tsc = d["x"], d["off"][lo:hi], d["length"][lo:hi], int(d["tsc"])
    base = int(off[0])
    xs = np.ascontiguousarray(x[base:int(off[-1] + length[-1])])
    off = (off - base).astype(np.int32)
    run = (lambda: r.rach_batch(xs, off, length)) if kind == "rach" else (lambda: r.normal_batch(xs, off, length, tsc))
    if kind == "config5":
        run = lambda: r.eq_batch(xs, off, length, tsc, 3.0, float(d["energy_thresh"]), int(d["max_toa"]))
    run()                                                # warm (page in, allocator)
    start.wait()
    for _ in range(reps):
        run()
    done.wait()


def main():
    path, P, seconds = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    import refbind
    d = np.load(path)
    kind = kind_of(d)
    r = refbind.Ref(int(d["sps"]), variant="52m" if kind == "config5" else "")
    if kind == "config4":
        iq = d["iq"]
        S = iq.shape[0]
        P = min(P, S)
        eq = "equalize" in d.files and int(d["equalize"]) != 0
        freqs, CW = (d["freqs"], int(d["rate_factor"])) if "freqs" in d.files else (None, 1)
        config4_stream(r, iq[0], d["lpf"], int(d["sps"]), int(d["tsc"]), eq, freqs, CW)        # warm: the whole stream once
        t0 = time.perf_counter()
        nb1 = config4_stream(r, iq[0], d["lpf"], int(d["sps"]), int(d["tsc"]), eq, freqs, CW)
        per_burst = (time.perf_counter() - t0) / nb1
        B = nb1 * S                                          # bursts per pass over all streams
        units, what = S, ("%d %sstreams x %d chunks (%d bursts): unUSRPify + %spolyphaseResampleVector chunk by chunk + slicing + "
                          "%s" % (S, "wideband " if freqs is not None else "", iq.shape[1] // (864 * CW), B,
                                  "per carrier (%d) frequencyShift + " % len(freqs) if freqs is not None else "",
                                  "energyDetect + analyzeTrafficBurst(requestChannel) + designDFE + equalizeBurst"
                                  if eq else "analyzeTrafficBurst + demodulateBurst"))
    else:
        B = len(d["off"])
        # single-process calibration on the first 512 (normal) / 128 (access) bursts
        n1 = min(512 if kind == "normal" else 128, B)
        off1, len1 = d["off"][:n1], d["length"][:n1]
        x1 = np.ascontiguousarray(d["x"][:int(off1[-1] + len1[-1])])     # (an .npz member is re-read on every access)
        run1 = (lambda: r.rach_batch(x1, off1, len1)) if kind == "rach" else (lambda: r.normal_batch(x1, off1, len1, int(d["tsc"])))
        if kind == "config5":
            run1 = lambda: r.eq_batch(x1, off1, len1, int(d["tsc"]), 3.0, float(d["energy_thresh"]), int(d["max_toa"]))
        run1()
        t0 = time.perf_counter()
        run1()
        per_burst = (time.perf_counter() - t0) / n1
        chain = {"rach": "detectRACHBurst + demodulateBurst", "normal": "analyzeTrafficBurst + demodulateBurst",
                 "config5": "energyDetect + analyzeTrafficBurst(requestChannel, maxTOA) + designDFE + equalizeBurst"}[kind]
        units, what = B, "the first %d bursts of the GPU batch (%s" % (B, chain)
    per_pass = per_burst * (B / P)
    reps = max(1, int(seconds / max(per_pass, 1e-4)))
    ctx = mp.get_context("fork")
    start, done = ctx.Barrier(P + 1), ctx.Barrier(P + 1)
    edges = [units * i // P for i in range(P + 1)]
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
                      "sample": ("%d passes over %s" % (reps, what)) + (" of " if kind != "config4" else "; ") +
                                "%s/sigProcLib.cpp compiled in place, oracle/_ref/libref_sigproc%s.so, %d processes, %.1f s%s"
                                % ("Transceiver52M" if kind == "config5" else "Transceiver", "52m" if kind == "config5" else "", P, tt,
                                   ")" if kind != "config4" else "")}))


if __name__ == "__main__":
    main()
