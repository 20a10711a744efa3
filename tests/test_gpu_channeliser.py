"""The multi-ARFCN channeliser (trxsig_rxfe_create_wideband / trxsig_rxfe_push_wideband): one wideband
int16 stream at 8 x 400 kS/s carrying four ARFCNs -> per carrier frequencyShift + polyphaseResampleVector(260 : 768) behind
the 192 x 8-sample history, fused in one kernel -> the front end's receive buffers -> burst slicing -> TSC detect + demod.
There is no such component in the reference (it runs one radio and one Transceiver per ARFCN): the oracle is the
reference's own two primitives applied per carrier (oracle/sigproc_oracle.c, pinned on the compiled reference by
tests/golden/extras.npz and resample.npz), with the mixer's phase chain restarted every 64 samples from
(float) fmod(64 b (double) freq, 2 pi) as include/trxsig_frontend.h defines it.  The resampled streams are compared value
for value; the bursts cut from them are detected and demodulated to the bits that were sent."""
import math

import numpy as np
import pytest

import _pkg
import oraclebind
import synth

pytestmark = pytest.mark.gpu

CW = 8                                              # wideband rate = 8 x 400 kS/s = 3.2 MS/s
SPS = 4


def make_wideband(Sw, offsets_hz, nbursts, tsc, seed):
    """int16 wideband streams: every carrier back-to-back normal bursts (157-156-156-156) at its frequency offset."""
    from scipy.signal import resample_poly
    rng = np.random.default_rng(seed)
    fs = 400e3 * CW
    streams, bits_all = [], []
    for w in range(Sw):
        tot = None
        bits_w = []
        for f in offsets_hz:
            bits = synth.normal_bits(rng, nbursts, tsc)
            base = synth.modulate(bits, SPS)
            guard = np.where(np.arange(nbursts) % 4 == 0, 9, 8)
            sig = np.concatenate([base[i, :(148 + guard[i]) * SPS] for i in range(nbursts)])
            wide = resample_poly(sig, 96 * CW, 65 * SPS) * (1500.0 * np.exp(2j * np.pi * rng.uniform()))
            n = np.arange(wide.size)
            wide = wide * np.exp(2j * np.pi * f / fs * n)
            tot = wide if tot is None else tot[:min(tot.size, wide.size)] + wide[:min(tot.size, wide.size)]
            bits_w.append(bits)
        tot = tot + (rng.standard_normal(tot.size) + 1j * rng.standard_normal(tot.size)) * 10.0
        iq = np.empty((tot.size, 2), np.int16)
        iq[:, 0] = np.clip(np.round(tot.imag), -32768, 32767)             # the radio delivers Q first
        iq[:, 1] = np.clip(np.round(tot.real), -32768, 32767)
        streams.append(iq); bits_all.append(bits_w)
    chunk = 864 * CW
    n = min(len(s) for s in streams) // chunk * chunk
    return np.stack([s[:n] for s in streams]), n // chunk, bits_all


def running_sum_chain(o, xc, freq, hist, chunk, nchunks, P, Q, lpf):
    """The same per-carrier chain with frequencyShift used as the reference wrote it: one call per chunk, the phase a running
    float sum, the next chunk started at the previous call's final phase reduced mod 2 pi."""
    start = np.float32(0.0)
    h = np.zeros(hist, np.complex64); out = []
    for c in range(nchunks):
        z, fin = o.frequency_shift(xc[c * chunk:(c + 1) * chunk], np.float32(freq), start)
        start = np.float32(math.fmod(float(fin), 6.283185307179586))
        win = np.concatenate([h, z])
        out.append(o.polyphase_resample(win, P, Q, lpf)[2 * P:]); h = win[-hist:]
    return np.concatenate(out)


def test_channeliser_equals_reference_primitives_per_carrier():
    import torch
    assert torch.cuda.is_available()
    pkg = _pkg.load()
    from openbts_ttsou_amd.frontend import RxFrontEnd
    Sw, tsc = 2, 3
    offsets = (-600e3, -200e3, 200e3, 600e3)
    fs = 400e3 * CW
    freqs = np.float32([-2.0 * np.pi * f / fs for f in offsets])         # bring each carrier down to 0
    P, Q = 65 * SPS, 96 * CW
    lpf = synth.design_lpf(8001, P, beta=6.0, cutoff=0.09)                # ~145 kHz at the 3.2 MS/s input rate
    iq, nchunks, bits_all = make_wideband(Sw, offsets, 24, tsc, seed=3)
    assert nchunks >= 5
    nchunks = 5
    ctx = pkg.TrxSig(SPS, 0); ctx.use_torch_stream()
    fe = RxFrontEnd(ctx, Sw, lpf, max_chunks=3, carrier_freq=freqs, rate_factor=CW)
    C = len(offsets)
    S = Sw * C
    o = oraclebind.Oracle(SPS)
    chunk, hist = 864 * CW, 192 * CW
    # oracle: mix the whole stream per carrier (the block convention makes it independent of the chunking), then pullBuffer
    xc = (iq[:, :nchunks * chunk, 1].astype(np.float32) + 1j * iq[:, :nchunks * chunk, 0].astype(np.float32)).astype(np.complex64)
    rcv = []
    for w in range(Sw):
        for k in range(C):
            z = o.mix_down(xc[w], hist, freqs[k])             # the stream's first sample is raw sample `hist` of the mixer's count
            h = np.zeros(hist, np.complex64); out = []
            for c in range(nchunks):
                win = np.concatenate([h, z[c * chunk:(c + 1) * chunk]])
                y = o.polyphase_resample(win, P, Q, lpf)
                out.append(y[2 * P:]); h = win[-hist:]
            rcv.append(np.concatenate(out))
    # device: pushes of 2 + 3 chunks
    d_iq = torch.from_numpy(np.ascontiguousarray(iq[:, :nchunks * chunk])).cuda()
    got = [np.zeros(0, np.complex64) for _ in range(S)]
    det_bits = 0
    c = 0
    tn = 0
    for k in (2, 3):
        fe.push_wideband(d_iq[:, c * chunk:(c + k) * chunk]); c += k
        popped = fe.pop_bursts()
        assert popped is not None
        x, off, length, tnv = popped
        nb = off.numel() // S
        xh = x.cpu().numpy().view(np.complex64).ravel(); offh = off.cpu().numpy(); lenh = length.cpu().numpy()
        for s in range(S):
            n = int(lenh[s * nb:(s + 1) * nb].sum())
            got[s] = np.concatenate([got[s], xh[offh[s * nb]:offh[s * nb] + n]])
        B = S * nb
        flags = torch.zeros(B, dtype=torch.uint8, device="cuda"); amp = torch.zeros(B, 2, device="cuda")
        toa = torch.zeros(B, device="cuda"); soft = torch.zeros(B, 148, device="cuda")
        ctx.detect_demod_normal(x, off, length, tsc, flags, amp, toa, soft, energy_thresh=50.0)
        torch.cuda.synchronize()
        fl = flags.cpu().numpy(); sf = soft.cpu().numpy()
        for s in range(S):
            for j in range(nb):
                i = s * nb + j
                gidx = tn + j                                           # the j-th burst cut since the start = burst gidx sent
                if gidx >= 1 and (fl[i] & pkg.F_DETECT):                 # (the first burst loses its head to the filter delay)
                    want = bits_all[s // C][s % C][gidx]
                    assert np.array_equal((sf[i] > 0.5).astype(np.uint8), want), (s, gidx)
                    det_bits += 1
        tn += nb
    for s in range(S):
        n = got[s].size
        assert n > 4 * 624
        assert np.array_equal(got[s], rcv[s][:n]), "stream %d (wideband %d, carrier %d)" % (s, s // C, s % C)
    assert det_bits >= S * (tn - 2) * 0.9, (det_bits, S, tn)             # nearly every complete burst came back right
    # against frequencyShift used with its running float phase: same bursts, soft bits close, hard bits equal.  The two differ by
    # a slowly drifting rotation (the running sum loses precision as it grows), most of which the per-burst amplitude estimate
    # takes out again.
    worst = 0.0
    for s in (0, S - 1):
        ref = running_sum_chain(o, xc[s // C], freqs[s % C], hist, chunk, nchunks, P, Q, lpf)
        pos = 628
        for j in range(1, 6):
            n = 624 + 4 * (j % 4 == 0)
            a = o.analyze_traffic(got[s][pos:pos + n], tsc, 3.0); b = o.analyze_traffic(ref[pos:pos + n], tsc, 3.0)
            assert a["ok"] and b["ok"]
            sa = o.demodulate(got[s][pos:pos + n], a["amp"], a["toa"])[:148]; sb = o.demodulate(ref[pos:pos + n], b["amp"], b["toa"])[:148]
            assert np.array_equal(sa > 0.5, sb > 0.5)
            worst = max(worst, float(np.abs(sa - sb).max()))
            pos += n
    print("channeliser vs running-sum frequencyShift: worst soft-bit difference %.2e" % worst)
    assert worst < 0.1                                                  # (measured: 3e-2 -- the running sum's own drift, not ours)



@pytest.mark.parametrize("offsets", [(-600e3, -200e3, 200e3, 600e3), (-1400e3, -1000e3, -600e3, -200e3, 200e3, 600e3, 1000e3, 1400e3)])
def test_shared_filter_form_against_the_per_carrier_form(offsets):
    """trxsig_rxfe_set_shared_filter: the carriers lie on the grid of sixteenths of the wideband rate, so ONE pass over the raw
    samples (sixteen partial sums of real taps, then sixteen complex multiply-adds per carrier) replaces the per-carrier mixers
    and filters.  Graded against the per-carrier form (the one pinned on the reference's primitives): resampled streams within
    1e-4 of the signal's scale, the same bursts detected, soft bits within 1e-4, hard bits identical -- over pushes of different
    sizes; frequencies off the grid and filters longer than 32 taps per output are refused."""
    import torch
    assert torch.cuda.is_available()
    pkg = _pkg.load()
    from openbts_ttsou_amd.frontend import RxFrontEnd
    Sw, tsc = 2, 3
    fs = 400e3 * CW
    freqs = np.float32([-2.0 * np.pi * f / fs for f in offsets])
    P = 65 * SPS
    lpf = synth.design_lpf(8001, P, beta=6.0, cutoff=0.09)
    iq, nchunks, bits_all = make_wideband(Sw, offsets, 30, tsc, seed=11)
    nchunks = min(nchunks, 6)
    C = len(offsets); S = Sw * C
    chunk = 864 * CW
    d_iq = torch.from_numpy(np.ascontiguousarray(iq[:, :nchunks * chunk])).cuda()
    ctx = pkg.TrxSig(SPS, 0); ctx.use_torch_stream()
    res = {}
    for shared in (False, True):
        fe = RxFrontEnd(ctx, Sw, lpf, max_chunks=3, carrier_freq=freqs, rate_factor=CW)
        if shared:
            fe.set_shared_filter(True)
        xs, softs, flags_all = [np.zeros(0, np.complex64) for _ in range(S)], [], []
        c = 0
        for k in (1, 3, 2)[:3]:
            k = min(k, nchunks - c)
            if k <= 0:
                break
            fe.push_wideband(d_iq[:, c * chunk:(c + k) * chunk]); c += k
            x, off, length, tnv = fe.pop_bursts()
            nb = off.numel() // S
            xh = x.cpu().numpy().view(np.complex64).ravel(); offh = off.cpu().numpy(); lenh = length.cpu().numpy()
            for s in range(S):
                n = int(lenh[s * nb:(s + 1) * nb].sum())
                xs[s] = np.concatenate([xs[s], xh[offh[s * nb]:offh[s * nb] + n]])
            B = S * nb
            flags = torch.zeros(B, dtype=torch.uint8, device="cuda"); amp = torch.zeros(B, 2, device="cuda")
            toa = torch.zeros(B, device="cuda"); soft = torch.zeros(B, 148, device="cuda")
            ctx.detect_demod_normal(x, off, length, tsc, flags, amp, toa, soft, energy_thresh=50.0)
            torch.cuda.synchronize()
            softs.append(soft.cpu().numpy()); flags_all.append(flags.cpu().numpy())
        res[shared] = (xs, np.concatenate(softs), np.concatenate(flags_all))
        fe.close()
    (xa, sa, fa), (xb, sb, fb) = res[False], res[True]
    scale = max(float(np.abs(x).max()) for x in xa)
    # The per-carrier form mixes with the carrier frequency AS A float32 (phase = n * (double) freq), the shared form with the
    # grid frequency 2 pi k / 16 itself: the float32 is off by up to 6e-8 rad per sample (a 0.03 Hz offset), which after 40,000
    # samples is a rotation of milliradians -- a property of the number handed in, not of either form.  Taken out (the rotation
    # dtheta * n at output o, n = o * 768 / 260 raw samples) the two agree to 1e-4 of the signal's scale and better.
    worst = raw_worst = 0.0
    for s_, (a, b) in enumerate(zip(xa, xb)):
        f32 = float(freqs[s_ % C])
        dth = f32 - 2.0 * np.pi * np.rint(f32 / (2.0 * np.pi / 16.0)) / 16.0
        n_raw = np.arange(a.size, dtype=np.float64) * (96.0 * CW) / (65.0 * SPS)
        raw_worst = max(raw_worst, float(np.abs(a - b).max()) / scale)
        ac = a * np.exp(-1j * dth * n_raw)
        ac = ac * np.exp(-1j * np.angle(np.vdot(b, ac)))      # (and the constant part: n counts from the mixer's sample 0, filter delay included)
        worst = max(worst, float(np.abs(ac - b).max()) / scale)
    print("shared-filter form vs per-carrier form: worst sample difference %.2e of the signal's scale (%.2e before the float32 "
          "carrier frequencies' own drift is taken out)" % (worst, raw_worst))
    assert worst < 1e-4 and raw_worst < 5e-3
    assert np.array_equal(fa, fb)
    det = (fa & pkg.F_DETECT) != 0
    assert det.sum() > 0.7 * det.size
    assert float(np.abs(sa[det] - sb[det]).max()) < 1e-4
    assert np.array_equal(sa[det] > 0.5, sb[det] > 0.5)
    # refusals
    off_grid = RxFrontEnd(ctx, 1, lpf, max_chunks=1, carrier_freq=np.float32([0.1, 0.2]), rate_factor=CW)
    with pytest.raises(pkg.TrxSigError):
        off_grid.set_shared_filter(True)
    off_grid.close()
    long_lpf = RxFrontEnd(ctx, 1, synth.design_lpf(33 * P + 1, P, beta=6.0, cutoff=0.09), max_chunks=1, carrier_freq=freqs[:2], rate_factor=CW)
    with pytest.raises(pkg.TrxSigError):
        long_lpf.set_shared_filter(True)
    long_lpf.close()
