#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref) -- build container only.

    make -C oracle ref && python oracle/gen_golden.py

Every array written here is either an input (seeded synthetic complex baseband,
bit patterns, filter coefficients) or the output the reference's own compiled
functions produced for that input.  The reference cannot travel to the GPU box;
these vectors can.  TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refbind  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, **kw):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **kw)
    print("wrote %s (%.1f KB)" % (name, os.path.getsize(path) / 1024.0))


def awgn(rng, n, sigma):
    return (sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).astype(np.complex64)


def normal_bits(rng, tscb, tsc):
    """148-bit normal burst: 3 tail, 57 data, steal, 26 TSC at bit 61, steal, 57 data, 3 tail
    (GSM/GSML1FEC.cpp:726)."""
    bits = rng.integers(0, 2, 148).astype(np.int8)
    bits[:3] = 0
    bits[-3:] = 0
    bits[61:87] = tscb[tsc]
    return bits


def rach_bits(rng, rachb):
    """8 bits 01010101 + 41-bit sync + 36 payload + 63 zeros (sigProcLibTest.cpp:38-47 layout)."""
    bits = np.zeros(148, np.int8)
    bits[:8] = [0, 1, 0, 1, 0, 1, 0, 1]
    bits[8:49] = rachb
    bits[49:85] = rng.integers(0, 2, 36)
    return bits


def pack(bursts):
    lens = np.array([len(b) for b in bursts], np.int32)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    return np.concatenate(bursts).astype(np.complex64), off, lens


def gen_tables():
    kw = {}
    for sps in (1, 2, 4):
        r = refbind.Ref(sps)
        c, s = r.trig_tables()
        rot, rev = r.rotation()
        mids = [r.midamble(t) for t in range(8)]
        rs, rt, rg = r.rach()
        p = "sps%d_" % sps
        kw.update({p + "rot": rot, p + "rev": rev, p + "pulse": r.pulse().real.copy(),
                   p + "mid": np.stack([m[0] for m in mids]),
                   p + "mid_toa": np.array([m[1] for m in mids], np.float32),
                   p + "mid_gain": np.array([m[2] for m in mids], np.complex64),
                   p + "rach": rs, p + "rach_toa": np.float32(rt), p + "rach_gain": np.complex64(rg)})
        kw["cosT"] = c
        kw["sinT"] = s
    tscb, dummy, rachb = r.gsm_bits()
    kw.update(training_sequence=tscb, dummy_burst=dummy, rach_synch=rachb)
    r52 = refbind.Ref(1, "52m")
    kw["sps1_52m_mid_toa"] = np.array([r52.midamble(t)[1] for t in range(8)], np.float32)
    # sinc(M_PI_F * d) on the 1/512 grid d in [-11, 11]: every argument interpolatePoint /
    # delayVector can form on the path; float32(pi) * float32(d)
    r = refbind.Ref(4)
    k = np.arange(-11 * 512, 11 * 512 + 1)
    d = (k / 512.0).astype(np.float32)
    arg = (np.float32(np.pi) * d).astype(np.float32)
    kw["sinc_grid"] = np.array([r.sinc(a) for a in arg], np.float32)
    xs = np.random.default_rng(7).uniform(-40, 40, 4096).astype(np.float32)
    kw["trig_x"] = xs
    kw["trig_sin"] = np.array([r.sinLookup(x) for x in xs], np.float32)
    kw["trig_cos"] = np.array([r.cosLookup(x) for x in xs], np.float32)
    kw["trig_sinc"] = np.array([r.sinc(x) for x in xs], np.float32)
    save("tables.npz", **kw)


def gen_modulate():
    rng = np.random.default_rng(11)
    kw = {}
    for sps in (1, 4):
        r = refbind.Ref(sps)
        tscb, dummy, rachb = r.gsm_bits()
        bits = [dummy, normal_bits(rng, tscb, 3), rach_bits(rng, rachb), np.zeros(148, np.int8),
                np.ones(148, np.int8), normal_bits(rng, tscb, 7)]
        guards = [8, 9, 8, 9, 8, 9]
        outs = [r.modulate(b, g) for b, g in zip(bits, guards)]
        x, off, lens = pack(outs)
        kw.update({"sps%d_bits" % sps: np.stack(bits), "sps%d_guard" % sps: np.array(guards, np.int32),
                   "sps%d_x" % sps: x, "sps%d_off" % sps: off, "sps%d_len" % sps: lens})
    save("modulate.npz", **kw)


def synth_normal(r, rng, tscb, it, sps, sigmas=(0.0, 0.1, 0.3)):
    tsc = it % 8
    bits = normal_bits(rng, tscb, tsc)
    x = r.modulate(bits, 8 + (it % 4 == 0))
    A = rng.uniform(300, 3000) * np.exp(2j * np.pi * rng.uniform())
    x = (x * np.complex64(A)).astype(np.complex64)
    d = np.float32(rng.uniform(-1.5, 1.5))
    x = r.delay_vector(x, d)
    x = (x + awgn(rng, x.size, sigmas[it % len(sigmas)] * abs(A))).astype(np.complex64)
    return x, bits, tsc, np.complex64(A), d


def gen_normal(sps, n, name):
    r = refbind.Ref(sps)
    rng = np.random.default_rng(100 + sps)
    tscb, _, _ = r.gsm_bits()
    X, BITS, TSC, A, D = [], [], [], [], []
    for it in range(n):
        x, bits, tsc, a, d = synth_normal(r, rng, tscb, it, sps)
        # edge cases the reference handles (SURVEY 8a' items 2, 12): silence, noise only,
        # midamble far out of the search window, clipped amplitude
        if it == n - 1: x = np.zeros_like(x)
        if it == n - 2: x = awgn(rng, x.size, 50.0)
        if it == n - 3: x = r.delay_vector(x, 30.0 * sps)
        if it == n - 4: x = (x * np.complex64(1e-3)).astype(np.complex64)
        X.append(x); BITS.append(bits); TSC.append(tsc); A.append(a); D.append(d)
    ok, amp, toa, ptm = [], [], [], []
    soft = np.zeros((n, 157), np.float32); nsoft = []
    chan = np.zeros((n, 6 * sps), np.complex64); chan_len = []; chan_off = np.zeros(n, np.float32)
    en_ok = []; en_pwr = []
    for it, x in enumerate(X):
        ra = r.analyze_traffic(x, TSC[it], 3.0, req_chan=True)
        rb = r.analyze_traffic(x, TSC[it], 3.0, req_chan=False)
        assert ra["ok"] == rb["ok"] and ra["amp"] == rb["amp"] and ra["toa"] == rb["toa"]
        ok.append(ra["ok"]); amp.append(ra["amp"]); toa.append(ra["toa"])
        chan_len.append(len(ra.get("chan", [])))
        if "chan" in ra:
            chan[it] = ra["chan"]; chan_off[it] = ra["chan_off"]
        # demodulate whenever amp != 0 (also for undetected bursts: exercises the demod path)
        if ra["amp"] != 0:
            s = r.demodulate(x, ra["amp"], ra["toa"])
            soft[it, :len(s)] = s; nsoft.append(len(s))
        else:
            nsoft.append(0)
        e = r.energy_detect(x, 20 * sps, 250.0)
        en_ok.append(e[0]); en_pwr.append(e[1])
    x, off, lens = pack(X)
    save(name, sps=sps, x=x, off=off, len=lens, bits=np.stack(BITS), tsc=np.array(TSC, np.int32),
         tx_amp=np.array(A), tx_delay=np.array(D, np.float32),
         ok=np.array(ok, np.uint8), amp=np.array(amp, np.complex64), toa=np.array(toa, np.float32),
         soft=soft, nsoft=np.array(nsoft, np.int32), chan=chan, chan_len=np.array(chan_len, np.int32),
         chan_off=chan_off, energy_ok=np.array(en_ok, np.uint8), energy_pwr=np.array(en_pwr, np.float32),
         energy_thresh=np.float32(250.0))


def gen_rach(sps, n, name):
    r = refbind.Ref(sps)
    rng = np.random.default_rng(200 + sps)
    _, _, rachb = r.gsm_bits()
    X, BITS, D = [], [], []
    for it in range(n):
        bits = rach_bits(rng, rachb)
        x = r.modulate(bits, 8 + (it % 4 == 0))
        A = rng.uniform(300, 3000) * np.exp(2j * np.pi * rng.uniform())
        x = (x * np.complex64(A)).astype(np.complex64)
        d = np.float32(rng.integers(0, 61) * sps + rng.uniform())
        x = r.delay_vector(x, d)
        x = (x + awgn(rng, x.size, (0.0, 0.1, 0.3)[it % 3] * abs(A))).astype(np.complex64)
        if it == n - 1: x = np.zeros_like(x)
        if it == n - 2: x = awgn(rng, x.size, 50.0)
        if it == n - 3: x = r.delay_vector(x, 100.0 * sps)   # valley window runs off the end
        X.append(x); BITS.append(bits); D.append(d)
    ok, amp, toa = [], [], []
    soft = np.zeros((n, 157), np.float32); nsoft = []
    for it, x in enumerate(X):
        rr = r.detect_rach(x, 5.0)
        ok.append(rr["ok"]); amp.append(rr["amp"]); toa.append(rr["toa"])
        if rr["amp"] != 0:
            s = r.demodulate(x, rr["amp"], rr["toa"])
            soft[it, :len(s)] = s; nsoft.append(len(s))
        else:
            nsoft.append(0)
    x, off, lens = pack(X)
    save(name, sps=sps, x=x, off=off, len=lens, bits=np.stack(BITS), tx_delay=np.array(D, np.float32),
         ok=np.array(ok, np.uint8), amp=np.array(amp, np.complex64), toa=np.array(toa, np.float32),
         soft=soft, nsoft=np.array(nsoft, np.int32))


def gen_primitives():
    rng = np.random.default_rng(300)
    r = refbind.Ref(4)

    def cn(n):
        return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    kw = {}
    i = 0
    for span in range(5):
        for (ar, br) in [(0, 0), (1, 0), (0, 1), (1, 1)]:
            for (na, nb) in [(40, 7), (40, 8), (7, 40), (1, 1)]:
                a = cn(na); b = cn(nb)
                if ar: a = a.real.astype(np.complex64)
                if br: b = b.real.astype(np.complex64)
                kw["conv%d_a" % i] = a; kw["conv%d_b" % i] = b
                kw["conv%d_meta" % i] = np.array([span, ar, br], np.int32)
                kw["conv%d_conv" % i] = r.convolve(a, b, span, ar, br)
                kw["conv%d_corr" % i] = r.correlate(a, b, span, ar, br)
                i += 1
    kw["nconv"] = np.int32(i)
    nd = 40
    xs, ds, ys = [], [], []
    for k in range(nd):
        n = int(rng.integers(5, 300)); x = cn(n); d = np.float32(rng.uniform(-8, 8))
        if k % 8 == 0: d = np.float32(np.round(d))
        if k % 8 == 1: d = np.float32(np.round(d) + 0.005)
        xs.append(x); ds.append(d); ys.append(r.delay_vector(x, d))
    x, off, lens = pack(xs)
    kw.update(delay_x=x, delay_off=off, delay_len=lens, delay_d=np.array(ds, np.float32), delay_y=np.concatenate(ys))
    ixs = np.array([rng.uniform(-3, lens[k] + 3) for k in range(nd)], np.float32)
    kw["interp_ix"] = ixs
    kw["interp_y"] = np.array([r.interpolate_point(xs[k], ixs[k]) for k in range(nd)], np.complex64)
    pk = [r.peak_detect(xs[k]) for k in range(nd)]
    kw["peak_val"] = np.array([p[0] for p in pk], np.complex64)
    kw["peak_idx"] = np.array([p[1] for p in pk], np.float32)
    kw["peak_avg"] = np.array([p[2] for p in pk], np.float32)
    pz = r.peak_detect(np.zeros(40, np.complex64))
    kw["peak_zero"] = np.array([pz[0].real, pz[0].imag, pz[1], pz[2]], np.float32)
    save("primitives.npz", **kw)


def gen_extras():
    """The part of sigProcLib.h the burst path never calls (dB, dBinv, vectorNorm2 / vectorPower, frequencyShift, sinc,
    addVector, offsetVector, gaussianNoise with a fixed srand seed, resampleVector, convolve's ABSSYM form), from the
    compiled reference."""
    rng = np.random.default_rng(700)
    r = refbind.Ref(1)

    def cn(n, scale=1.0):
        return (scale * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    kw = {}
    xs = np.concatenate([np.float32([0.0, -1.0, 1.0, 1.5, 0.5, 0.25, 1e-3, 3e-7, 1e-19, 0.99999994, 0.1]),
                         (10.0 ** rng.uniform(-20, 0.2, 40)).astype(np.float32)])
    kw["db_x"] = xs; kw["db_y"] = np.array([r.dB(v) for v in xs], np.float32)
    xi = np.concatenate([np.float32([0.0, 1.0, -200.0, -250.0, -3.0, -0.5, -10.7, -199.5, -12.0, -24.0]),
                         rng.uniform(-210, 2, 40).astype(np.float32)])
    kw["dbinv_x"] = xi; kw["dbinv_y"] = np.array([r.dBinv(v) for v in xi], np.float32)
    sx = np.concatenate([np.float32([0.0, 0.01, -0.01, 0.0099, 3.1415927, -3.1415927, 40.0, -77.7]),
                         rng.uniform(-35, 35, 60).astype(np.float32)])
    kw["sinc_x"] = sx; kw["sinc_y"] = np.array([r.sinc(v) for v in sx], np.float32)
    vec = [cn(n, 100.0) for n in (1, 2, 63, 64, 65, 157, 628, 1056)]
    x, off, lens = pack(vec)
    kw.update(vec_x=x, vec_off=off, vec_len=lens, vec_norm2=np.array([r.vector_norm2(v) for v in vec], np.float32),
              vec_power=np.array([r.vector_power(v) for v in vec], np.float32))
    # frequencyShift: complex and real-only inputs, phases that wrap both ways
    fs = []
    for k, v in enumerate(vec):
        freq = np.float32([0.0, 0.48 * np.pi, -0.3, 2.5, -6.9, 0.013, 1.0, -0.77][k])
        start = np.float32([0.0, 1.0, -2.0, 12.5, -40.0, 0.5, 3.1415927, 100.0][k])
        ro = k % 3 == 2
        vv = v.real.astype(np.complex64) if ro else v
        y, fin = r.frequency_shift(vv, freq, start, ro)
        fs.append((vv, freq, start, ro, y, fin))
    kw.update(fshift_x=np.concatenate([f[0] for f in fs]), fshift_freq=np.array([f[1] for f in fs], np.float32),
              fshift_start=np.array([f[2] for f in fs], np.float32), fshift_real=np.array([f[3] for f in fs], np.uint8),
              fshift_y=np.concatenate([f[4] for f in fs]), fshift_final=np.array([f[5] for f in fs], np.float32))
    # addVector: the shorter operand bounds the sum
    a = cn(157, 50.0); b1 = cn(157, 50.0); b2 = cn(40, 50.0); b3 = cn(300, 50.0)
    kw.update(add_x=a, add_y1=b1, add_y2=b2, add_y3=b3, add_r1=r.add_vector(a, b1), add_r2=r.add_vector(a, b2), add_r3=r.add_vector(a, b3))
    o = np.complex64(3.25 - 1.5j)
    ar = a.real.astype(np.complex64)
    kw.update(offset=o, offset_c=r.offset_vector(a, o, False), offset_xr=ar, offset_r=r.offset_vector(ar, o, True))
    # gaussianNoise after srand(seed): the reference's rand() draw order (two draws per sample, more after a zero)
    for seed, n, var, mean in ((1, 149, 0.001 / np.sqrt(np.float32(2)), 0j), (12345, 64, 2.5, 1.0 - 2.0j)):
        kw["noise%d" % seed] = r.gaussian_noise(seed, n, var, mean)
        kw["noise%d_arg" % seed] = np.array([n, var, mean.real, mean.imag], np.float64)
    # resampleVector as it behaves: every interpolated value lands in element 0
    rs = cn(50, 10.0)
    for k, ef in enumerate((1.0, 1.5, 2.37, 4.0)):
        kw["rsv%d" % k] = r.resample_vector(rs, ef, 7 + 1j)
    kw.update(rsv_x=rs, rsv_factor=np.float32([1.0, 1.5, 2.37, 4.0]), rsv_end=np.complex64(7 + 1j))
    # convolve with an ABSSYM filter (START_ONLY / NO_DELAY: the spans on which the reference stays inside its operands)
    i = 0
    for span in (refbind.START_ONLY, refbind.NO_DELAY):
        for nb in (4, 5, 8, 9, 21):
            aa = cn(60); bb = cn(nb)
            kw["sym%d_a" % i] = aa; kw["sym%d_b" % i] = bb; kw["sym%d_span" % i] = np.int32(span)
            kw["sym%d_y" % i] = r.convolve(aa, bb, span, abssym=True)
            i += 1
    kw["nsym"] = np.int32(i)
    save("extras.npz", **kw)


def gen_resample():
    rng = np.random.default_rng(400)
    r = refbind.Ref(4)
    rcv, snd = r.lpf_raw()
    snd961 = np.concatenate([snd, [0.0]]).astype(np.float32)   # tap[960] = 0
    kw = dict(rcvLPF_651_raw=rcv, sendLPF_961_raw=snd961)
    kw["lpf651_gain260"] = r.create_lpf651(260.0)
    kw["lpf651_gain65"] = r.create_lpf651(65.0)
    kw["lpf651_gain96"] = r.create_lpf651(96.0)

    def cn(n, scale=1.0):
        return (scale * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    # RX direction as the radio interface drives it (radioInterface.cpp:230-246): 192 history +
    # 864 new samples at 400 kS/s -> 65*sps/96; int16-valued samples
    for sps in (1, 4):
        x = np.round(cn(1056, 3000.0).view(np.float32)).view(np.complex64)
        kw["rx%d_x" % sps] = x
        kw["rx%d_y651" % sps] = r.polyphase_resample(x, 65 * sps, 96, kw["lpf651_gain260"] if sps == 4 else kw["lpf651_gain65"])
    # the RX call site really uses the 961 table with gain 65*sps (radioInterface.cpp:233)
    g = np.float32(260.0)
    s = snd961.astype(np.float64).sum()
    lpf961 = (snd961 * np.float32(g / s)).astype(np.float32)
    kw["lpf961_gain260"] = lpf961
    kw["rx4_y961"] = r.polyphase_resample(kw["rx4_x"], 260, 96, lpf961)
    # TX direction (radioInterface.cpp:137-144): 96 : 65*sps with the 651 table, gain 96
    x = cn(2 * 260 + 625 * 4)
    kw["tx4_x"] = x
    kw["tx4_y"] = r.polyphase_resample(x, 96, 260, kw["lpf651_gain96"])
    save("resample.npz", **kw)


def gen_dfe():
    """Config 5: Transceiver52M pullRadioVector flow at sps=1 (Transceiver52M/Transceiver.cpp:268-407):
    energyDetect(stride 4) -> analyzeTrafficBurst(maxTOA=4, requestChannel) -> scale chan by 1/amp ->
    designDFE(Nf=7) -> scale burst by 1/amp -> equalizeBurst(TOA - chanOffset).  Inputs are integers
    with |v| <= 2048 (fp16-exact)."""
    r = refbind.Ref(1, "52m")
    rng = np.random.default_rng(500)
    tscb, _, _ = r.gsm_bits()
    n = 64
    X, BITS, TSC = [], [], []
    for it in range(n):
        tsc = it % 8
        bits = normal_bits(rng, tscb, tsc)
        x = r.modulate(bits, 8 + (it % 4 == 0))
        A = rng.uniform(200, 900) * np.exp(2j * np.pi * rng.uniform())
        x = (x * np.complex64(A)).astype(np.complex64)
        x = r.delay_vector(x, np.float32(rng.uniform(-1.0, 1.0)))
        ch = np.array([1, 0.4 + 0.2j, 0], np.complex64) if it % 2 else np.array([1, 0, 0], np.complex64)
        x = r.convolve(x, ch, refbind.START_ONLY)
        x = (x + awgn(rng, x.size, (0.01, 0.05, 0.1)[it % 3] * abs(A))).astype(np.complex64)
        x = np.clip(np.round(x.view(np.float32)), -2048, 2048).astype(np.float32).view(np.complex64)
        X.append(x); BITS.append(bits); TSC.append(tsc)
    ok, amp, toa, en = [], [], [], []
    chan = np.zeros((n, 6), np.complex64); chan_off = np.zeros(n, np.float32); chan_len = []
    W = np.zeros((n, 7), np.complex64); Bf = np.zeros((n, 5), np.complex64)
    soft = np.zeros((n, 157), np.float32); snr = np.zeros(n, np.float32)
    thr = np.float32(10.0)
    for it, x in enumerate(X):
        en.append(r.energy_detect(x, 20, thr))
        ra = r.analyze_traffic(x, TSC[it], 3.0, req_chan=True, max_toa=4)
        ok.append(ra["ok"]); amp.append(ra["amp"]); toa.append(ra["toa"])
        chan_len.append(len(ra.get("chan", [])))
        if ra["ok"]:
            a = ra["amp"]
            # amplitude.norm2()/(mEnergyThreshold*mEnergyThreshold+1.0): float / double -> double -> float
            # (Transceiver52M/Transceiver.cpp:336)
            n2a = np.float32(np.float32(a.imag * a.imag) + np.float32(a.real * a.real))
            snr[it] = np.float32(np.float64(n2a) / (np.float64(np.float32(thr * thr)) + 1.0))
            chan[it] = ra["chan"]; chan_off[it] = ra["chan_off"]
            # scaleVector(*channelResp, complex(1,0)/amplitude): 1/amp formed by the reference itself
            n2 = np.float32(a.imag * a.imag + a.real * a.real)
            inv = complex(np.float32(a.real / n2), np.float32(-a.imag / n2))
            chn = r.scale_vector(ra["chan"], inv)
            w, b = r.design_dfe(chn, float(snr[it]), 7)
            W[it] = w; Bf[it, :len(b)] = b
            xs = r.scale_vector(x, inv)
            s = r.equalize(xs, np.float32(ra["toa"] - ra["chan_off"]), w, b)
            soft[it, :len(s)] = s
    x, off, lens = pack(X)
    save("dfe_52m_sps1.npz", x=x, off=off, len=lens, bits=np.stack(BITS), tsc=np.array(TSC, np.int32),
         energy_thresh=thr, energy_ok=np.array([e[0] for e in en], np.uint8),
         energy_pwr=np.array([e[1] for e in en], np.float32),
         ok=np.array(ok, np.uint8), amp=np.array(amp, np.complex64), toa=np.array(toa, np.float32),
         chan=chan, chan_len=np.array(chan_len, np.int32), chan_off=chan_off, snr=snr, w=W, b=Bf, soft=soft)
    nerr = sum(int(((soft[i, :148] > 0.5) != BITS[i]).sum()) for i in range(n) if ok[i])
    print("  dfe: detected %d/%d, hard-bit errors after DFE: %d" % (sum(ok), n, nerr))


def gen_config1():
    """BASELINE config 1: the sigProcLibTest.cpp call sequence (Transceiver/sigProcLibTest.cpp:29-181)
    at sps=1 on the Transceiver/ variant, CommSig->BitVector, SoftSig->SoftVector.
    The 961-tap LPF is built here with tap[960]=0 instead of createLPF(...,961,...) (SURVEY a21:
    that call reads one float past sendLPF_961[])."""
    sps = 1
    r = refbind.Ref(sps)
    rng = np.random.default_rng(600)
    tscb, _, rachb = r.gsm_bits()
    kw = {}
    # RACH leg (sigProcLibTest.cpp:38-53): "01010101" + sync + 99 zeros, guard 9
    rb = np.zeros(148, np.int8); rb[:8] = [0, 1, 0, 1, 0, 1, 0, 1]; rb[8:49] = rachb
    xr = r.modulate(rb, 9)
    rr = r.detect_rach(xr, 5.0)
    kw.update(rach_bits=rb, rach_x=xr, rach_ok=np.uint8(rr["ok"]), rach_amp=rr["amp"], rach_toa=rr["toa"])
    # normal-burst leg (sigProcLibTest.cpp:76-167)
    pay = np.array([int(c) for c in "0000101010100111110010101010010110101110011000111001101010000"], np.int8)
    bits = np.concatenate([pay, tscb[0], pay]).astype(np.int8)
    x = r.modulate(bits, 0)                                     # :84-85 guard 0
    rcv, snd = r.lpf_raw()
    snd961 = np.concatenate([snd, [0.0]]).astype(np.float32)
    lpf_tx = r.create_lpf651(96.0)                              # :91  createLPF(.,651,P=96)
    ssum = snd961.astype(np.float64).sum()
    lpf_rx = (snd961 * np.float32(np.float32(65.0) / ssum)).astype(np.float32)   # :98 createLPF(.,961,P=65)
    up = r.polyphase_resample(x, 96, 65, lpf_tx)                # :105-108
    dn = r.polyphase_resample(up, 65, 96, lpf_rx)               # :112-113
    dl = r.delay_vector(dn, np.float32(6.932))                  # :125
    ch = np.array([9000.0, np.float32(0.4) * np.float32(9000.0), 0, np.float32(-1.2) * 0], np.complex64)  # :133-137
    rx = r.convolve(dl, ch, refbind.NO_DELAY)                   # :139
    ra = r.analyze_traffic(rx, 0, 8.0, req_chan=True)           # :146 (before the noise is added)
    noise_pwr = 0.001 / float(np.sqrt(np.float32(2)))           # :143 double noisePwr = 0.001/sqrtf(2)
    noise = r.gaussian_noise(1, rx.size, np.float32(noise_pwr))  # :144 after srand(1), the seed a C program starts with
    rxn = r.add_vector(rx, noise)                               # :147
    kw.update(autocorr=r.correlate(dn, xr, refbind.NO_DELAY), energy=r.vector_norm2(up), noise=noise)   # :118, :122
    soft = r.demodulate(rxn, ra["amp"], ra["toa"])              # :152
    kw.update(bits=bits, mod=x, up=up, dn=dn, delayed=dl, rx=rx, rx_noisy=rxn, ok=np.uint8(ra["ok"]),
              amp=ra["amp"], toa=ra["toa"], chan=ra.get("chan", np.zeros(0, np.complex64)),
              chan_off=np.float32(ra.get("chan_off", 0)), soft=soft, lpf_tx=lpf_tx, lpf_rx=lpf_rx)
    if ra["ok"]:
        snr = np.float32(1.0 / noise_pwr)
        w, b = r.design_dfe(ra["chan"], float(snr), 7)          # :159
        eq = r.equalize(rxn, np.float32(ra["toa"] - ra["chan_off"]), w, b)   # :164
        kw.update(dfe_snr=snr, dfe_w=w, dfe_b=b, eq_soft=eq)
        print("  config1: rach ok=%s toa=%.4f; TSC detect ok, TOA %.4f amp %s; slicer bit errors %d, DFE bit errors %d" % (
            rr["ok"], rr["toa"], ra["toa"], ra["amp"], int(((soft[:148] > 0.5) != bits).sum()),
            int(((eq[:148] > 0.5) != bits).sum())))
    save("config1_loopback.npz", **kw)


def gen_gsm_time():
    """GSM::Time ordering and arithmetic (GSM/GSMCommon.h:327-455, GSMCommon.cpp:161-176) from the compiled reference: the
    part of Transceiver's host logic (priority queue order, stale-burst test, frame differences) that CAN be pinned --
    Transceiver.cpp itself needs the USRP headers and TRXManager.cpp libosip2, neither builds here."""
    r = refbind.Ref(1)
    H = 2048 * 26 * 51
    rng = np.random.default_rng(20261004)
    edge = [0, 1, 2, H // 2 - 1, H // 2, H // 2 + 1, H - 2, H - 1, 25, 26, 50, 51, 101, 102]
    fns = np.array(edge + list(rng.integers(0, H, 200)), np.int64)
    A = [(int(f), int(t)) for f in fns for t in (0, 3, 7)]
    pairs = [(A[i], A[j]) for i in rng.integers(0, len(A), 1500) for j in [int(rng.integers(0, len(A)))]]
    pairs += [((f, t), (f2, t2)) for f in edge for f2 in edge for (t, t2) in ((0, 0), (2, 5), (7, 1))]
    g = lambda op, x, y=(0, 0), s=0: refbind.gsm_time(r.lib, op, x, y, s)
    steps = rng.integers(0, 9, len(pairs)).astype(np.int32)
    fsteps = rng.integers(-3000, 3000, len(pairs)).astype(np.int32)
    save("gsm_time.npz",
         a=np.array([p[0] for p in pairs], np.int32), b=np.array([p[1] for p in pairs], np.int32),
         less=np.array([g(0, x, y) for x, y in pairs], np.int32), greater=np.array([g(1, x, y) for x, y in pairs], np.int32),
         equal=np.array([g(2, x, y) for x, y in pairs], np.int32), minus=np.array([g(3, x, y) for x, y in pairs], np.int32),
         tn_step=steps, inc_tn=np.array([g(5, x, s=s) for (x, _), s in zip(pairs, steps)], np.int32),
         dec_tn=np.array([g(6, x, s=s) for (x, _), s in zip(pairs, steps)], np.int32),
         fn_step=fsteps, add_fn=np.array([g(7, x, s=s) for (x, _), s in zip(pairs, fsteps)], np.int32),
         plus=np.array([g(8, x, y) for x, y in pairs], np.int32))


BITVECTORTEST_MC = ("000000000000111100000000000001110000011100001101000011000000000000000111000011110000100100001010"
                    "000010100000101000001010000010100000010000000000000000000000000000000000000000000000001100001111"
                    "000000000000000000000000000000000000000000000000000010010000101000001010000010100000101000001010"
                    "000001000000000000000000000000110000111100000000000001110000101000001100000001000000000000")


def gen_fec():
    """L1 FEC soft decode from the REAL reference BitVector / ViterbiR2O4 / Parity code
    (oracle/_ref/libref_fec.so): XCCH blocks, RACH bursts, bare Viterbi runs incl. the input string of the
    reference's own CommonLibs/BitVectorTest.cpp:72, parity / syndrome words."""
    import reffec
    r = reffec.RefFec()
    rng = np.random.default_rng(20260104)

    def wire(v):                        # Transceiver.cpp:669 + TRXManager.cpp:231
        q = np.round(v.astype(np.float64) * 255.0).astype(np.int64).astype(np.uint8)
        return (q.astype(np.float32) / np.float32(256.0)).astype(np.float32)

    kw = {}
    # --- XCCH: 96 blocks at several noise levels; every third through the UDP quantisation; some hopeless
    nb = 96
    d = rng.integers(0, 2, (nb, 184)).astype(np.uint8)
    hard = np.stack([r.xcch_encode(d[i]) for i in range(nb)])                       # [nb,4,114]
    sig = np.array([0.0, 0.05, 0.15, 0.25, 0.35, 0.5, 0.8, 2.0])[np.arange(nb) % 8]
    soft = np.clip(hard * 0.8 + 0.1 + rng.normal(0, 1, hard.shape) * sig[:, None, None], 0, 1).astype(np.float32)
    soft[5] = 0.5; soft[6] = 0.0; soft[7] = 1.0                                     # unknown / all-zero / all-one
    soft[8, 2] = 0.5                                                                # one burst missing (fec:626-628)
    wq = (np.arange(nb) % 3) == 0
    soft[wq] = wire(soft[wq])
    res = [r.xcch_decode(soft[i]) for i in range(nb)]
    kw.update(xcch_d=d, xcch_hard=hard.astype(np.uint8), xcch_soft=soft, xcch_wire=wq,
              xcch_ok=np.array([x["ok"] for x in res]), xcch_u=np.stack([x["u"] for x in res]),
              xcch_dout=np.stack([x["d"] for x in res]),
              xcch_syndrome=np.array([x["syndrome"] for x in res], np.uint64))
    # --- RACH: valid access bursts for a known BSIC, noisy, and garbage
    nr = 96
    ra = rng.integers(0, 256, nr)
    bsic = rng.integers(0, 64, nr)
    e_hard = np.zeros((nr, 36), np.uint8)
    for i in range(nr):
        u = np.zeros(18, np.uint8)
        u[:8] = r.lsb8msb(np.array([(ra[i] >> (7 - k)) & 1 for k in range(8)], np.uint8))
        chk = r.parity(0x06f, 6, 8, u[:8])
        sent = (~(chk ^ int(bsic[i]))) & 0x3f                    # decoder: bsic = (~sent ^ chk) & 0x3f (fec:490-493)
        u[8:14] = [(sent >> (5 - k)) & 1 for k in range(6)]
        e_hard[i] = r.encode(u)
    sg = np.array([0.0, 0.1, 0.2, 0.3, 0.45, 1.0])[np.arange(nr) % 6]
    e = np.clip(e_hard * 0.8 + 0.1 + rng.normal(0, 1, e_hard.shape) * sg[:, None], 0, 1).astype(np.float32)
    e[4] = rng.random(36); e[5] = 0.5
    rr = [r.rach_decode(e[i]) for i in range(nr)]
    kw.update(rach_ra=ra, rach_bsic=bsic, rach_e=e, rach_u=np.stack([x["u"] for x in rr]),
              rach_tail_ok=np.array([x["tail_ok"] for x in rr]), rach_bsic_out=np.array([x["bsic"] for x in rr]),
              rach_ra_out=np.array([x["ra"] for x in rr]))
    # --- TCH/FS: class-1 Viterbi + class-2 slicing + 3-bit parity, from the reference's decodeTCH steps
    nt = 64
    td = rng.integers(0, 2, (nt, 260)).astype(np.uint8)
    tc = np.stack([r.tch_encode(td[i]) for i in range(nt)]).astype(np.float32)
    tsg = np.array([0.0, 0.1, 0.2, 0.3, 0.45, 0.7, 1.5, 0.0])[np.arange(nt) % 8]
    tsoft = np.clip(tc * 0.8 + 0.1 + rng.normal(0, 1, tc.shape) * tsg[:, None], 0, 1).astype(np.float32)
    tsoft[7] = 0.5
    tw = (np.arange(nt) % 3) == 1
    tsoft[tw] = wire(tsoft[tw])
    tr = [r.tch_decode(tsoft[i]) for i in range(nt)]
    kw.update(tch_d=td, tch_soft=tsoft, tch_good=np.array([x["good"] for x in tr]),
              tch_u=np.stack([x["u"] for x in tr]), tch_dout=np.stack([x["d"] for x in tr]))
    # --- bare Viterbi runs
    mc = np.array([int(c) for c in BITVECTORTEST_MC], np.uint8)
    kw.update(kat_c=mc, kat_u=r.soft_decode(mc.astype(np.float32), len(mc) // 2))
    lens = [2, 4, 36, 50, 100, 378, 456]
    for n in lens:
        sv = rng.random(n).astype(np.float32)
        kw["vit%d_in" % n] = sv
        kw["vit%d_out" % n] = r.soft_decode(sv, n // 2)
        bits = rng.integers(0, 2, n // 2).astype(np.uint8)
        kw["enc%d_in" % n] = bits
        kw["enc%d_out" % n] = r.encode(bits)
    # --- parity / syndrome words
    pb = rng.integers(0, 2, (16, 224)).astype(np.uint8)
    kw.update(par_bits=pb,
              par_xcch=np.array([r.parity(0x10004820009, 40, 224, pb[i, :184]) for i in range(16)], np.uint64),
              syn_xcch=np.array([r.syndrome(0x10004820009, 40, 224, pb[i]) for i in range(16)], np.uint64),
              par_rach=np.array([r.parity(0x06f, 6, 8, pb[i, :8]) for i in range(16)], np.uint64))
    save("fec.npz", **kw)


if __name__ == "__main__":
    if len(sys.argv) > 1:            # regenerate selected files only, e.g. `gen_golden.py dfe`
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    if not refbind.available():
        sys.exit("oracle/_ref not built: run `make -C oracle ref` in the build container")
    gen_tables()
    gen_modulate()
    gen_normal(4, 96, "normal_sps4.npz")
    gen_normal(1, 64, "normal_sps1.npz")
    gen_rach(4, 48, "rach_sps4.npz")
    gen_rach(1, 32, "rach_sps1.npz")
    gen_primitives()
    gen_extras()
    gen_resample()
    gen_dfe()
    gen_config1()
    gen_gsm_time()
    gen_fec()
