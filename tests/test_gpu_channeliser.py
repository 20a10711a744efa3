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
