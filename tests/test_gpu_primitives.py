"""GPU parity of sigProcLib.h's free-standing vector primitives (csrc/trxsig_prim.hip: convolve / correlate in every span
type and real/complex form, delayVector, interpolatePoint, peakDetect, scaleVector, GMSKRotate / GMSKReverseRotate,
vectorSlicer, decimateVector) through the C-ABI: the golden vectors captured from the real reference
(tests/golden/primitives.npz) and seeded random / ragged / degenerate inputs against the pinned oracle.  Bit-exact."""
import numpy as np
import pytest

import _pkg
import oraclebind
from util import assert_beq, assert_veq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _pkg.load()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = {s: pkg.TrxSig(s, 0) for s in (1, 4)}
    for v in c.values():
        v.use_torch_stream()
    return c


def cn(rng, n, scale=1.0):
    return (scale * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)


def test_golden_primitives(ctx, golden):
    g = golden("primitives.npz")
    t = ctx[4]
    for i in range(int(g["nconv"])):
        span, ar, br = (int(v) for v in g["conv%d_meta" % i])
        a, b = g["conv%d_a" % i], g["conv%d_b" % i]
        assert_beq(t.convolve_host(a, b, span, ar, br), g["conv%d_conv" % i], "convolve %d" % i)
        assert_beq(t.convolve_host(a, b, span, ar, br, correlate=True), g["conv%d_corr" % i], "correlate %d" % i)
    pos = 0
    for k, (off, n) in enumerate(zip(g["delay_off"], g["delay_len"])):
        x = g["delay_x"][off:off + n]
        assert_beq(t.delay_vector_host(x, g["delay_d"][k]), g["delay_y"][pos:pos + n], "delayVector %d" % k)
        pos += n
        assert t.interpolate_point_host(x, g["interp_ix"][k]) == g["interp_y"][k], k
        v, ix, avg = t.peak_detect_host(x)
        assert v == g["peak_val"][k] and ix == g["peak_idx"][k] and avg == g["peak_avg"][k], k
    v, ix, avg = t.peak_detect_host(np.zeros(40, np.complex64))      # all-zero: maxIndex -1 (sigProcLib.cpp:669)
    assert_beq(np.array([v.real, v.imag, ix, avg], np.float32), g["peak_zero"])


@pytest.mark.parametrize("sps", [1, 4])
def test_random_primitives_vs_oracle(ctx, sps):
    o = oraclebind.Oracle(sps)
    t = ctx[sps]
    rng = np.random.default_rng(77 + sps)
    for trial in range(24):
        na, nb = int(rng.integers(1, 700)), int(rng.integers(1, 170))
        a, b = cn(rng, na), cn(rng, nb)
        span = trial % 5
        ar, br = bool(trial & 1), bool(trial & 2)
        if ar: a = a.real.astype(np.complex64)
        if br: b = b.real.astype(np.complex64)
        assert_beq(t.convolve_host(a, b, span, ar, br), o.convolve(a, b, span, ar, br), "convolve span %d" % span)
        assert_beq(t.convolve_host(a, b, span, ar, br, correlate=True), o.correlate(a, b, span, ar, br), "correlate span %d" % span)
    # the windowed CUSTOM span of the 52 MHz variant (Transceiver52M/sigProcLib.cpp:301-304)
    a, b = cn(rng, 156), cn(rng, 16)
    assert_beq(t.convolve_host(a, b, 5, cust_start=58, cust_len=9), o.convolve(a, b, 5, start=58, length=9), "CUSTOM span")
    for trial in range(40):
        n = int(rng.integers(2, 700))
        x = cn(rng, n, 100.0)
        d = np.float32(rng.uniform(-30, 30))
        if trial % 5 == 0: d = np.float32(np.round(d))                     # integer delay: no filtering (:582)
        if trial % 5 == 1: d = np.float32(np.round(d) + 0.009)             # |frac| <= 1e-2: still none
        if trial % 7 == 0: d = np.float32(d * 40)                          # shifted right out of the vector
        assert_beq(t.delay_vector_host(x, d), o.delay_vector(x, d), "delayVector n %d d %g" % (n, d))
        ix = np.float32(rng.uniform(-25, n + 25))                          # incl. the ranges the clamps of :643-646 open up
        assert t.interpolate_point_host(x, ix) == o.interpolate_point(x, ix), (n, ix)
        if n >= 3:
            v, i, avg = t.peak_detect_host(x)
            ov, oi, oavg = o.peak_detect(x)
            assert v == ov and i == oi and avg == oavg, (n, v, ov, i, oi, avg, oavg)
        s = np.complex64(complex(rng.normal(), rng.normal()))
        assert_beq(t.elementwise_host(0, x, s), o.scale_vector(x, s), "scaleVector")
        m = min(n, 157 * sps)
        assert_beq(t.elementwise_host(1, x[:m]), o.gmsk_rotate(x[:m]), "GMSKRotate")
        assert_beq(t.elementwise_host(2, x[:m]), o.gmsk_rotate(x[:m], reverse=True), "GMSKReverseRotate")
    # energyDetect: any window, both strides (the 52 MHz variant steps four samples)
    o52 = oraclebind.Oracle(1, variant52m=True)
    for n, win in ((628, 80), (157, 20), (50, 200), (300, 1), (97, 64), (130, 65)):
        x = cn(rng, n, 30.0)
        for thr in (10.0, 40.0, 80.0):
            ok, av = t.energy_detect_host(x, win, thr)
            ook, oav = o.energy_detect(x, win, thr)
            assert ok == ook and av == oav, (n, win, thr, av, oav)
            if 4 * min(win, n) <= n:
                ok, av = t.energy_detect_host(x, win, thr, step=4)
                ook, oav = o52.energy_detect(x, win, thr)
                assert ok == ook and av == oav, ("52M", n, win, thr, av, oav)
    # peaks at the ends and flat inputs: the bisection's clamped interpolation ranges
    for x in (np.r_[np.complex64(50), cn(rng, 30)], np.r_[cn(rng, 30), np.complex64(50)], np.full(20, 3 + 4j, np.complex64),
              np.array([1, 2], np.complex64)):
        x = np.ascontiguousarray(x, np.complex64)
        v, i, avg = t.peak_detect_host(x)
        ov, oi, oavg = o.peak_detect(x)
        assert v == ov and i == oi and avg == oavg, (x.size, v, ov, i, oi)
    # vectorSlicer (:507-519) and decimateVector (:1039-1053): closed forms
    x = cn(rng, 333, 1.5)
    want = np.clip((0.5 * (x.real.astype(np.float64) + np.float64(np.float32(1.0)))).astype(np.float32), 0, 1)
    got = t.elementwise_host(3, x)
    assert_beq(got.real.copy(), want, "vectorSlicer"); assert not got.imag.any()
    for f in (2, 3, 4):
        n = 120 * f
        assert_beq(t.decimate_host(x[:n] if n <= x.size else np.resize(x, n), f), (x[:n] if n <= x.size else np.resize(x, n))[::f], "decimateVector")


def test_batch_forms_ragged(pkg, ctx):
    """The device-pointer batch entry points on ragged batches: same values as vector by vector through the oracle."""
    import torch
    t = ctx[4]; o = oraclebind.Oracle(4)
    rng = np.random.default_rng(5)
    B = 37
    lens = rng.integers(1, 640, B).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    x = cn(rng, int(lens.sum()), 10.0)
    b = cn(rng, 41)
    dev = torch.device("cuda:0")
    dx = torch.from_numpy(x.view(np.float32).copy()).to(dev); doff = torch.from_numpy(off).to(dev); dlen = torch.from_numpy(lens).to(dev)
    db = torch.from_numpy(b.view(np.float32).copy()).to(dev)
    L = t.L
    for span in range(5):
        olen = np.array([L.trxsig_convolve_out_len(int(n), 41, span, 0) for n in lens], np.int32)
        ooff = np.concatenate([[0], np.cumsum(olen)[:-1]]).astype(np.int32)
        dout = torch.zeros(int(olen.sum()) * 2, device=dev); dooff = torch.from_numpy(ooff).to(dev)
        t._chk(L.trxsig_convolve_batch(t.h, dx.data_ptr(), doff.data_ptr(), dlen.data_ptr(), B, int(lens.max()), db.data_ptr(), 41, span,
                                       0, 1, 0, 0, dout.data_ptr(), dooff.data_ptr()), "convolve_batch")
        got = dout.cpu().numpy().view(np.complex64)
        for i in range(B):
            assert_beq(got[ooff[i]:ooff[i] + olen[i]], o.correlate(x[off[i]:off[i] + lens[i]], b, span), "span %d vector %d" % (span, i))
    delay = rng.uniform(-20, 20, B).astype(np.float32)
    ddelay = torch.from_numpy(delay).to(dev); dy = torch.zeros_like(dx)
    t._chk(L.trxsig_delay_vector_batch(t.h, dx.data_ptr(), doff.data_ptr(), dlen.data_ptr(), B, ddelay.data_ptr(), 0, dy.data_ptr()), "delay")
    pk = torch.zeros(B, 2, device=dev); ix = torch.zeros(B, device=dev); av = torch.zeros(B, device=dev)
    t._chk(L.trxsig_peak_detect_batch(t.h, dx.data_ptr(), doff.data_ptr(), dlen.data_ptr(), B, pk.data_ptr(), ix.data_ptr(), av.data_ptr()), "peak")
    y = dy.cpu().numpy().view(np.complex64); pk = pk.cpu().numpy().view(np.complex64).ravel(); ix = ix.cpu().numpy(); av = av.cpu().numpy()
    for i in range(B):
        xi = x[off[i]:off[i] + lens[i]]
        assert_beq(y[off[i]:off[i] + lens[i]], o.delay_vector(xi, delay[i]), "delay %d" % i)
        if lens[i] >= 2:
            ov, oi, oavg = o.peak_detect(xi)
            assert pk[i] == ov and ix[i] == oi and av[i] == oavg, i
    # bad arguments are refused, not launched
    with pytest.raises(pkg.TrxSigError):
        t._chk(L.trxsig_delay_vector_batch(t.h, dx.data_ptr(), doff.data_ptr(), dlen.data_ptr(), B, ddelay.data_ptr(), 0, dx.data_ptr()), "in place")
    with pytest.raises(pkg.TrxSigError):
        t._chk(L.trxsig_convolve_batch(t.h, dx.data_ptr(), doff.data_ptr(), dlen.data_ptr(), B, int(lens.max()), db.data_ptr(), 41, 9,
                                       0, 0, 0, 0, dy.data_ptr(), doff.data_ptr()), "span 9")


def test_unreducible_delay_and_index_return_promptly(pkg, ctx):
    """The hang class of round 3 (a TOA of -1e9 sent the table sinc's subtract-one range reduction into a loop that never
    ended): delays / indices the reference's loop (sigProcLib.cpp:163-188 under :573-616 and :639-659) cannot reduce are
    refused by the host forms (TRXSIG_EINVAL, no launch) and give zeros in the batch forms; large but reducible ones still
    equal the oracle bit for bit (the device reduction is the loop's closed form); hostile SAMPLES (NaN / inf) cannot keep
    peakDetect running either.  Everything returns within seconds."""
    import time
    import torch
    t = ctx[4]; o = oraclebind.Oracle(4)
    rng = np.random.default_rng(404)
    x = cn(rng, 300, 5.0)
    t0 = time.time()
    for bad in (1e10, -1e9, 3.0e7, -2.0e7, float("inf"), float("-inf"), float("nan")):
        with pytest.raises(pkg.TrxSigError):
            t.delay_vector_host(x, bad)
        with pytest.raises(pkg.TrxSigError):
            t.interpolate_point_host(x, bad)
    # large, reducible: the closed form is the loop
    for ix in (-3000.25, 5000.5, -1.0e5, 123456.75, 150.3, -0.7, 298.9, 299.5):
        assert t.interpolate_point_host(x, ix) == o.interpolate_point(x, np.float32(ix)), ix
    for d in (-999999.5, 4096.25, 16777216.0, -16777216.0, 299.5, -300.125):
        assert_beq(t.delay_vector_host(x, d), o.delay_vector(x, np.float32(d)), "delay %r" % d)
    # batch forms: hostile entries -> zeros, the others untouched by their neighbours
    B = 12
    lens = np.full(B, 300, np.int32); off = (np.arange(B) * 300).astype(np.int32)
    xs = cn(rng, 300 * B, 3.0)
    vals = np.array([1e10, -1e9, np.inf, -np.inf, np.nan, 3.5, -2.25, 1e30, -1e30, 0.0, 2.0e7, 7.125], np.float32)
    dev = torch.device("cuda:0")
    dx = torch.from_numpy(xs.view(np.float32).copy()).to(dev); doff = torch.from_numpy(off).to(dev); dlen = torch.from_numpy(lens).to(dev)
    dv = torch.from_numpy(vals).to(dev); dy = torch.full_like(dx, 7.0); dp = torch.full((B, 2), 7.0, device=dev)
    L = t.L
    t._chk(L.trxsig_delay_vector_batch(t.h, dx.data_ptr(), doff.data_ptr(), dlen.data_ptr(), B, dv.data_ptr(), 0, dy.data_ptr()), "delay")
    t._chk(L.trxsig_interpolate_point_batch(t.h, dx.data_ptr(), doff.data_ptr(), dlen.data_ptr(), B, dv.data_ptr(), 0, dp.data_ptr()), "interp")
    torch.cuda.synchronize()
    y = dy.cpu().numpy().view(np.complex64); p = dp.cpu().numpy().view(np.complex64).ravel()
    for i in range(B):
        xi = xs[off[i]:off[i] + 300]
        if np.isfinite(vals[i]) and abs(vals[i]) <= 16777216.0:
            assert_beq(y[off[i]:off[i] + 300], o.delay_vector(xi, vals[i]), "batch delay %d" % i)
            assert p[i] == o.interpolate_point(xi, vals[i]), i
        else:
            assert not y[off[i]:off[i] + 300].any() and p[i] == 0, i
    # hostile samples
    for hostile in (np.nan, np.inf, -np.inf, 3.0e38):
        xh = x.copy(); xh[100] = hostile; xh[7] = complex(0, hostile)
        t.peak_detect_host(xh)
        t.delay_vector_host(xh, 0.37)
    assert time.time() - t0 < 30.0


def test_rest_of_the_sigproclib_surface(ctx, golden):
    """dB / dBinv, sinc, vectorNorm2 / vectorPower, frequencyShift, addVector, offsetVector, gaussianNoise (fixed srand seed),
    resampleVector and convolve's ABSSYM form through the C-ABI against tests/golden/extras.npz (captured from the compiled
    reference) and, on random inputs, the oracle.  Bit-exact."""
    g = golden("extras.npz")
    t = ctx[1]
    o = oraclebind.Oracle(1)
    assert_beq(np.array([t.db(v) for v in g["db_x"]], np.float32), g["db_y"], "dB")
    assert_beq(np.array([t.dbinv(v) for v in g["dbinv_x"]], np.float32), g["dbinv_y"], "dBinv")
    assert_beq(np.array([t.sinc_host(v) for v in g["sinc_x"]], np.float32), g["sinc_y"], "sinc")
    pos = 0
    for k, (off, n) in enumerate(zip(g["vec_off"], g["vec_len"])):
        v = g["vec_x"][off:off + n]
        e, p = t.vector_norm2_host(v)
        assert e == g["vec_norm2"][k] and p == g["vec_power"][k], k
        x = g["fshift_x"][pos:pos + n]
        y, fin = t.frequency_shift_host(x, g["fshift_freq"][k], g["fshift_start"][k], bool(g["fshift_real"][k]))
        assert_beq(y, g["fshift_y"][pos:pos + n], "frequencyShift %d" % k)
        assert fin == g["fshift_final"][k]
        pos += n
    for k in (1, 2, 3):
        assert_beq(t.add_vector_host(g["add_x"], g["add_y%d" % k]), g["add_r%d" % k], "addVector %d" % k)
    assert_beq(t.elementwise_host(4, g["add_x"], complex(g["offset"]), False), g["offset_c"], "offsetVector")
    assert_beq(t.elementwise_host(4, g["offset_xr"], complex(g["offset"]), True), g["offset_r"], "offsetVector real")
    for seed in (1, 12345):
        n, var, mr, mi = g["noise%d_arg" % seed]
        assert_beq(t.gaussian_noise_host(seed, int(n), np.float32(var), complex(mr, mi)), g["noise%d" % seed], "gaussianNoise %d" % seed)
    for k, ef in enumerate(g["rsv_factor"]):
        assert_beq(t.resample_linear_host(g["rsv_x"], ef, complex(g["rsv_end"])), g["rsv%d" % k], "resampleVector %d" % k)
    assert t.resample_linear_host(g["rsv_x"], 0.5) is None
    for i in range(int(g["nsym"])):
        assert_beq(t.convolve_host(g["sym%d_a" % i], g["sym%d_b" % i], int(g["sym%d_span" % i]), abssym=True), g["sym%d_y" % i], "ABSSYM %d" % i)
    # random inputs against the oracle, incl. every span of the ABSSYM form and phases that wrap many times
    rng = np.random.default_rng(5)
    for trial in range(16):
        n = int(rng.integers(1, 900))
        x = cn(rng, n, 100.0)
        assert t.vector_norm2_host(x) == (o.vector_norm2(x), o.vector_power(x))
        f, s0, ro = np.float32(rng.uniform(-7, 7)), np.float32(rng.uniform(-50, 50)), bool(trial & 1)
        xx = x.real.astype(np.complex64) if ro else x
        y, fin = t.frequency_shift_host(xx, f, s0, ro); yo, fo = o.frequency_shift(xx, f, s0, ro)
        assert_beq(y, yo, "frequencyShift random %d" % trial); assert fin == fo
        b = cn(rng, int(rng.integers(1, 30)))
        for span in range(5):
            assert_beq(t.convolve_host(x, b, span, abssym=True), o.convolve(x, b, span, abssym=True), "ABSSYM span %d" % span)
        ef = np.float32(rng.uniform(1, 5))
        assert_beq(t.resample_linear_host(x, ef, 1 - 2j), o.resample_vector(x, ef, 1 - 2j), "resampleVector random")
    with pytest.raises(Exception):
        t.frequency_shift_host(np.ones(100, np.complex64), 1000.0, 0.0)      # beyond +-25000 rad: the reference would not return
