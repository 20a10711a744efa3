"""A larger parity run than the test-suite's: GPU (through the C-ABI) against the CPU oracle, value-exact, on many
seeds -- normal bursts at every TSC and sps with noise levels up to "undetectable", access bursts, the 52M equaliser
leg.  Since round 5 every demodulated batch is ALSO run in the tolerance mode (trxsig_set_soft_mode): detection, amplitude and TOA must
stay value-exact, every hard bit identical, every soft bit within the guaranteed 7.4e-5 of the oracle's (the largest is printed).
Run on the GPU box:  python tools/parity_campaign.py [bursts-per-case]   (about a minute with 16 host cores)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch
import _pkg
import oraclebind
from util import GpuBatch, assert_veq
pkg = _pkg.load()
from openbts_ttsou_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
worst_tol = 0.0


def check_tolerance(t, run, want_soft, what):
    """The same batch in the tolerance mode: run() launches it and returns the results dictionary."""
    global worst_tol
    t.set_soft_mode(pkg.SOFT_TOLERANCE)
    r = run()
    t.set_soft_mode(pkg.SOFT_EXACT)
    d = np.abs(r["soft"].astype(np.float64) - want_soft.astype(np.float64))
    d = d[np.isfinite(d)]
    assert d.size == 0 or d.max() <= 7.4e-5, (what, d.max())
    assert np.array_equal(r["soft"] > 0.5, want_soft > 0.5), what
    worst_tol = max(worst_tol, float(d.max()) if d.size else 0.0)
    return r
NT = min(os.cpu_count() or 1, 16)
t0 = time.time()
total = 0
for sps in (1, 2, 4):
    t = pkg.TrxSig(sps, 0); t.use_torch_stream()
    o = oraclebind.Oracle(sps)
    for tsc in range(8):
        x, off, length, meta = synth.normal_batch(sps, N, tsc, seed=31337 + 100 * sps + tsc, sigmas=(0.0, 0.05, 0.2, 0.5, 1.0, 3.0),
                                                  max_delay=2.5)
        gb = GpuBatch(x, off, length)
        for ethr in (-1.0, 0.0):
            t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, hard=gb.hard, energy_thresh=ethr)
            r = gb.results()
            ok, amp, toa, soft = o.normal_batch(x, off, length, tsc, nthreads=NT)
            assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "detect sps%d tsc%d" % (sps, tsc))
            assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa"); assert_veq(r["soft"], soft, "soft")
            def again():
                t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, hard=gb.hard, energy_thresh=ethr)
                return gb.results()
            rt = check_tolerance(t, again, soft, "normal sps%d tsc%d" % (sps, tsc))
            assert_veq((rt["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "detect (tolerance mode)")
            assert_veq(rt["amp"], amp, "amp (tolerance mode)"); assert_veq(rt["toa"], toa, "toa (tolerance mode)")
        total += N
    x, off, length, meta = synth.rach_batch(sps, N // 2, seed=4242 + sps, sigmas=(0.0, 0.1, 0.3, 1.0, 4.0), max_delay_sym=90)
    gb = GpuBatch(x, off, length)
    t.detect_demod_rach(gb.x, gb.off, gb.len, gb.flags, gb.amp, gb.toa, gb.soft, energy_thresh=-1.0)
    r = gb.results()
    ok, amp, toa, soft = o.rach_batch(x, off, length, nthreads=NT)
    assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "rach detect sps%d" % sps)
    assert_veq(r["amp"], amp, "rach amp"); assert_veq(r["toa"], toa, "rach toa"); assert_veq(r["soft"], soft, "rach soft")
    def again_rach():
        t.detect_demod_rach(gb.x, gb.off, gb.len, gb.flags, gb.amp, gb.toa, gb.soft, energy_thresh=-1.0)
        return gb.results()
    rt = check_tolerance(t, again_rach, soft, "rach sps%d" % sps)
    assert_veq(rt["amp"], amp, "rach amp (tolerance mode)"); assert_veq(rt["toa"], toa, "rach toa (tolerance mode)")
    total += N // 2
    print("sps %d: 8 TSC x %d normal bursts (x2 energy gates) + %d access bursts identical  [%.0f s]" % (sps, N, N // 2, time.time() - t0),
          flush=True)
# the equaliser leg at one sample per symbol, classic and 52M windows, half of the bursts through a two-path channel
t = pkg.TrxSig(1, 0); t.use_torch_stream()
for variant52m in (False, True):
    o = oraclebind.Oracle(1, variant52m=variant52m)
    tsc, thr, Ne = 6, 10.0, max(N // 4, 256)
    x, off, length, meta = synth.normal_batch(1, Ne, tsc, seed=777 + variant52m, sigmas=(0.02, 0.1, 0.4, 1.5), max_delay=2.0)
    x = (x * np.float32(200)).astype(np.complex64)
    for i in range(1, Ne, 2):
        s = x[off[i]:off[i] + length[i]]
        s[1:] = s[1:] + np.complex64(0.4 + 0.2j) * s[:-1].copy()
    dx = torch.from_numpy(np.ascontiguousarray(x).view(np.float32)).cuda()
    doff = torch.from_numpy(off.astype(np.int32)).cuda(); dlen = torch.from_numpy(length.astype(np.int32)).cuda()
    fl = torch.zeros(Ne, dtype=torch.uint8, device="cuda"); am = torch.zeros(Ne, 2, device="cuda"); to = torch.zeros(Ne, device="cuda")
    so = torch.zeros(Ne, 157, device="cuda")
    t.equalize_normal(dx, doff, dlen, tsc, fl, am, to, so, energy_thresh=thr, variant52m=variant52m, max_toa=4, nsoft=156, soft_stride=157)
    torch.cuda.synchronize()
    flh = fl.cpu().numpy(); soh = so.cpu().numpy()
    for i in range(Ne):
        s = x[off[i]:off[i] + length[i]]
        ok_e, _ = o.energy_detect(s, 20, thr)
        assert bool(flh[i] & pkg.F_ENERGY) == ok_e, i
        a = o.analyze_traffic(s, tsc, 3.0, req_chan=True, max_toa=4) if ok_e else None
        det = bool(a and a["ok"])
        assert bool(flh[i] & pkg.F_DETECT) == det, i
        if det:
            amv = a["amp"]
            n2 = np.float32(np.float32(amv.imag * amv.imag) + np.float32(amv.real * amv.real))
            inv = complex(np.float32(amv.real / n2), np.float32(-amv.imag / n2))
            snr = np.float32(np.float64(n2) / (np.float64(np.float32(thr * thr)) + 1.0))
            w, b = o.design_dfe(o.scale_vector(a["chan"], inv), float(snr), 7)
            soft = o.equalize(o.scale_vector(s, inv), np.float32(a["toa"] - a["chan_off"]), w, b)
            assert np.array_equal(soh[i, :156], soft[:156]), i
    total += Ne
    print("equaliser leg (%s window): %d bursts identical  [%.0f s]" % ("52M" if variant52m else "classic", Ne, time.time() - t0), flush=True)
print("parity campaign: %d bursts, every output value-exact in the exact mode; tolerance mode on the same demodulated batches: detection, amplitude, "
      "TOA value-exact, hard bits identical, largest soft-bit difference %.3g (guaranteed <= 7.4e-5)" % (total, worst_tol))
