"""BASELINE config 4 end to end on the GPU: S independent int16 I/Q streams at 400 kS/s -> unUSRPify ->
polyphase resample 65*sps : 96 with the reference's LPF table -> 157/156/156/156 burst slicing ->
per-burst TSC detect + demod, against the same chain built from the CPU oracle.  Value-exact."""
import numpy as np
import pytest

import _pkg
import oraclebind
import synth
from util import assert_veq

pytestmark = pytest.mark.gpu


def make_streams(sps, S, nbursts, tsc, seed):
    """int16 I/Q streams at 400 kS/s carrying back-to-back normal bursts (157-156-156-156 symbols)."""
    from scipy.signal import resample_poly
    rng = np.random.default_rng(seed)
    out = []
    for s in range(S):
        bits = synth.normal_bits(rng, nbursts, tsc)
        base = synth.modulate(bits, sps)                         # [nb, 157*sps]
        guard = np.where(np.arange(nbursts) % 4 == 0, 9, 8)
        sig = np.concatenate([base[i, :(148 + guard[i]) * sps] for i in range(nbursts)])
        sig = sig * (2000.0 * np.exp(2j * np.pi * rng.uniform()))
        lo = resample_poly(sig, 96, 65 * sps)                    # to the radio's rate
        lo = lo + (rng.standard_normal(lo.size) + 1j * rng.standard_normal(lo.size)) * 20.0
        iq = np.empty((lo.size, 2), np.int16)
        iq[:, 0] = np.clip(np.round(lo.imag), -32768, 32767)     # the radio delivers Q first (I/Q flipped)
        iq[:, 1] = np.clip(np.round(lo.real), -32768, 32767)
        out.append(iq)
    n = min(len(o) for o in out) // 864 * 864
    return np.stack([o[:n] for o in out]), n // 864


def test_config4_pipeline(golden):
    import torch
    assert torch.cuda.is_available()
    pkg = _pkg.load()
    from openbts_ttsou_amd.frontend import RxFrontEnd, OUTCHUNK, OUTHISTORY
    sps, S, tsc = 4, 3, 5
    g = golden("resample.npz")
    lpf = g["lpf961_gain260"]                                   # createLPF(.,961,P=65*sps) as pullBuffer builds it
    iq, nchunks = make_streams(sps, S, 40, tsc, seed=11)
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    fe = RxFrontEnd(ctx, S, lpf)
    o = oraclebind.Oracle(sps)
    # oracle chain per stream
    hist = [np.zeros(OUTHISTORY, np.complex64) for _ in range(S)]
    rcv = [np.zeros(0, np.complex64) for _ in range(S)]
    tn_o = 0
    total = 0
    for c in range(nchunks):
        chunk = iq[:, c * OUTCHUNK:(c + 1) * OUTCHUNK]
        fe.push_chunk(torch.from_numpy(np.ascontiguousarray(chunk)).cuda())
        got = fe.pop_bursts()
        # --- oracle ---
        for s in range(S):
            cf = (chunk[s, :, 1].astype(np.float32) + 1j * chunk[s, :, 0].astype(np.float32)).astype(np.complex64)
            y = o.polyphase_resample(np.concatenate([hist[s], cf]), 65 * sps, 96, lpf)
            rcv[s] = np.concatenate([rcv[s], y[2 * 65 * sps:]])
            hist[s] = cf[-OUTHISTORY:]
        lens = []
        pos, tn = 0, tn_o
        while len(rcv[0]) - pos > (156 + (tn % 4 == 0)) * sps:
            n = (156 + (tn % 4 == 0)) * sps
            lens.append(n); pos += n; tn = (tn + 1) % 8
        tn_o = tn
        if not lens:
            assert got is None
            continue
        x, off, length, tnv = got
        nb = len(lens)
        assert off.numel() == S * nb
        # the resampled samples themselves are bit-identical
        xh = x.cpu().numpy().view(np.complex64).ravel()
        offh = off.cpu().numpy()
        for s in range(S):                                        # stream s's bursts are back to back from its first offset on
            assert_veq(xh[offh[s * nb]:offh[s * nb] + pos], rcv[s][:pos], "resampled stream %d chunk %d" % (s, c))
        B = S * nb
        flags = torch.zeros(B, dtype=torch.uint8, device="cuda"); amp = torch.zeros(B, 2, device="cuda")
        toa = torch.zeros(B, device="cuda"); soft = torch.zeros(B, 148, device="cuda")
        ctx.detect_demod_normal(x, off, length, tsc, flags, amp, toa, soft, energy_thresh=100.0)
        torch.cuda.synchronize()
        fl = flags.cpu().numpy(); a = amp.cpu().numpy().view(np.complex64).ravel(); t = toa.cpu().numpy()
        sf = soft.cpu().numpy()
        for s in range(S):
            p = 0
            for j, n in enumerate(lens):
                burst = rcv[s][p:p + n]; p += n
                i = s * nb + j
                eok, _ = o.energy_detect(burst, 20 * sps, 100.0)
                assert bool(fl[i] & pkg.F_ENERGY) == eok
                if not eok:
                    continue
                r = o.analyze_traffic(burst, tsc, 3.0)
                assert bool(fl[i] & pkg.F_DETECT) == r["ok"] and a[i] == r["amp"] and t[i] == r["toa"], (s, j)
                if r["ok"]:
                    assert_veq(sf[i], o.demodulate(burst, r["amp"], r["toa"])[:148], "soft %d/%d" % (s, j))
                    total += 1
            rcv[s] = rcv[s][pos:]
    assert total >= S * 20                                       # most complete slots carried a detectable burst


def test_multi_chunk_push_equals_chunk_by_chunk(golden):
    """K chunks in one push (one fused launch, every chunk behind the 192 samples before it) produce the same burst stream
    as K pushes of one chunk, and the buffers survive pushes without a pop in between up to their capacity."""
    import torch
    pkg = _pkg.load()
    from openbts_ttsou_amd.frontend import RxFrontEnd, OUTCHUNK
    sps, S, tsc = 4, 5, 1
    lpf = golden("resample.npz")["lpf961_gain260"]
    iq, nchunks = make_streams(sps, S, 60, tsc, seed=3)
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).cuda()

    def run(sizes):
        fe = RxFrontEnd(ctx, S, lpf, max_chunks=max(sizes), start_tn=3)
        out, c = [], 0
        for k in sizes:
            fe.push_chunk(d_iq[:, c * OUTCHUNK:(c + k) * OUTCHUNK]); c += k
            got = fe.pop_bursts()
            if got is None:
                continue
            x, off, length, tn = got
            xh = x.cpu().numpy().view(np.complex64).ravel(); off = off.cpu().numpy(); length = length.cpu().numpy()
            nb = len(off) // S
            out.append([(int(tn[j]), [xh[off[s * nb + j]:off[s * nb + j] + length[s * nb + j]].copy() for s in range(S)]) for j in range(nb)])
        return [b for grp in out for b in grp]
    one = run([1] * nchunks)
    many = run([7, 1, 5, nchunks - 13])
    assert len(one) == len(many) and len(one) > 50
    for (tn1, b1), (tn2, b2) in zip(one, many):
        assert tn1 == tn2
        for s in range(S):
            assert_veq(b1[s], b2[s], "burst stream %d" % s)
    fe = RxFrontEnd(ctx, S, lpf, max_chunks=2)
    fe.push_chunk(d_iq[:, :2 * OUTCHUNK])
    with pytest.raises(pkg.TrxSigError, match="full"):
        fe.push_chunk(d_iq[:, 2 * OUTCHUNK:4 * OUTCHUNK])


@pytest.mark.parametrize("sps,table,taps_per_output", [(1, "sendLPF_961_raw", 15), (1, "rcvLPF_651_raw", 11), (2, "sendLPF_961_raw", 8)])
def test_rx_resampler_with_many_taps_per_output(golden, sps, table, taps_per_output, request):
    """k_rx_resample<KQ> beyond four taps per output: the reference's own receive configuration is one sample per symbol, 65 : 96
    with its 961-tap table (createLPF(., 961, 65): radioInterface.cpp:230-234) = 15 taps per output.  The resampled streams
    equal the oracle's polyphaseResampleVector chunk by chunk behind the 192-sample history, bit for bit."""
    import torch
    assert torch.cuda.is_available()
    pkg = _pkg.load()
    from openbts_ttsou_amd.frontend import RxFrontEnd, OUTCHUNK, OUTHISTORY
    S = 3
    raw = golden("resample.npz")[table]
    h = pkg.TrxHost(sps, 0)
    lpf = h.create_lpf(raw, 65.0 * sps)
    h.close()
    assert (len(lpf) + 65 * sps - 1) // (65 * sps) == taps_per_output
    iq, nchunks = make_streams(sps, S, 12, 3, seed=21 + sps)
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    ctx.set_tuning(rxres_wpb=2)                                # two windows per workgroup (what a 128-stream push does) in this small case
    request.addfinalizer(lambda: pkg.TrxSig(sps, 0).set_tuning(rxres_wpb=0))      # (library-wide: back to the default)
    fe = RxFrontEnd(ctx, S, lpf, max_chunks=4)
    o = oraclebind.Oracle(sps)
    hist = [np.zeros(OUTHISTORY, np.complex64) for _ in range(S)]
    rcv = [np.zeros(0, np.complex64) for _ in range(S)]
    tn_o, checked, c = 0, 0, 0
    while c < nchunks:
        k = min(1 + (c % 3), nchunks - c)                     # pushes of 1, 2, 3 chunks
        chunk = iq[:, c * OUTCHUNK:(c + k) * OUTCHUNK]
        fe.push_chunk(torch.from_numpy(np.ascontiguousarray(chunk)).cuda())
        got = fe.pop_bursts()
        for s in range(S):
            for j in range(k):
                ch = chunk[s, j * OUTCHUNK:(j + 1) * OUTCHUNK]
                cf = (ch[:, 1].astype(np.float32) + 1j * ch[:, 0].astype(np.float32)).astype(np.complex64)
                y = o.polyphase_resample(np.concatenate([hist[s], cf]), 65 * sps, 96, lpf)
                rcv[s] = np.concatenate([rcv[s], y[2 * 65 * sps:]])
                hist[s] = cf[-OUTHISTORY:]
        c += k
        pos, tn = 0, tn_o
        while len(rcv[0]) - pos > (156 + (tn % 4 == 0)) * sps:
            pos += (156 + (tn % 4 == 0)) * sps; tn = (tn + 1) % 8
        if pos == 0:
            assert got is None
            continue
        tn_o = tn
        x, off, length, tnv = got
        nb = off.numel() // S
        xh = x.cpu().numpy().view(np.complex64).ravel(); offh = off.cpu().numpy()
        for s in range(S):
            assert_veq(xh[offh[s * nb]:offh[s * nb] + pos], rcv[s][:pos], "resampled stream %d after chunk %d" % (s, c))
            rcv[s] = rcv[s][pos:]
        checked += pos
    assert checked > 1200 * sps


@pytest.mark.parametrize("lpf_kind", ["reference_table", "designed"])
def test_fused_front_end_equals_push_pop_detect(golden, lpf_kind):
    """trxsig_rxfe_push_detect_demod_normal (the detectors compute their samples from the int16 chunks; no resampled stream
    in memory) against push + pop + trxsig_detect_demod_normal_batch on the same streams: pushes of 1, 2, 7, ... chunks, a
    start TN of 3, bursts straddling pushes.  Flags, amplitude, TOA, average power, soft and hard bits identical bit for bit."""
    import torch
    pkg = _pkg.load()
    from openbts_ttsou_amd import synth
    from openbts_ttsou_amd.frontend import RxFrontEnd, OUTCHUNK
    sps, S, tsc = 4, 6, 1
    lpf = golden("resample.npz")["lpf961_gain260"] if lpf_kind == "reference_table" else synth.design_lpf(961, 260)
    iq, nchunks = make_streams(sps, S, 130, tsc, seed=5)
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    sizes = [1, 1, 2, 7, 1, 5, 3]
    assert nchunks > sum(sizes) + 1
    sizes.append(nchunks - sum(sizes))
    dev = "cuda"

    def bufs(n):
        return dict(flags=torch.zeros(n, dtype=torch.uint8, device=dev), amp=torch.zeros(n, 2, device=dev), toa=torch.zeros(n, device=dev),
                    pwr=torch.zeros(n, device=dev), soft=torch.full((n, 148), -1.0, device=dev),
                    hard=torch.full((n, 148), 255, dtype=torch.uint8, device=dev))

    def unfused():
        fe = RxFrontEnd(ctx, S, lpf, max_chunks=max(sizes), start_tn=3)
        out, c = [], 0
        for k in sizes:
            fe.push_chunk(d_iq[:, c * OUTCHUNK:(c + k) * OUTCHUNK]); c += k
            r = fe.pop_raw()
            if r is None:
                out.append(None); continue
            ps, po, pl, tn, nb = r
            o = bufs(S * nb)
            ctx._chk(ctx.L.trxsig_detect_demod_normal_batch(ctx.h, ps, po, pl, S * nb, tsc, 3.0, 0.0, o["flags"].data_ptr(), o["amp"].data_ptr(),
                                                            o["toa"].data_ptr(), o["pwr"].data_ptr(), o["soft"].data_ptr(), o["hard"].data_ptr(),
                                                            148, 148), "detect_demod")
            torch.cuda.synchronize()
            out.append((tn, {k2: v.cpu().numpy() for k2, v in o.items()}))
        return out

    def fused():
        fe = RxFrontEnd(ctx, S, lpf, max_chunks=max(sizes), start_tn=3)
        out, c = [], 0
        for k in sizes:
            o = bufs(S * (2 + k * 4))
            nb, tn = fe.push_detect_demod(d_iq[:, c * OUTCHUNK:(c + k) * OUTCHUNK], tsc, o["flags"], o["amp"], o["toa"], o["soft"],
                                          avgpwr=o["pwr"], hard=o["hard"], nsoft=148, soft_stride=148)
            c += k
            torch.cuda.synchronize()
            out.append(None if nb == 0 else (tn, {k2: v.cpu().numpy()[:S * nb] for k2, v in o.items()}))
        return out

    a, b = unfused(), fused()
    ndet = ntot = 0
    for i, (ra, rb) in enumerate(zip(a, b)):
        assert (ra is None) == (rb is None), i
        if ra is None:
            continue
        assert np.array_equal(ra[0], rb[0]), ("tn", i)
        for k2 in ("flags", "amp", "toa", "pwr", "soft", "hard"):
            x, y = ra[1][k2], rb[1][k2]
            assert x.shape == y.shape and x.tobytes() == y.tobytes(), (k2, "push", i, sizes[i])
        ndet += int(((ra[1]["flags"] & pkg.F_DETECT) != 0).sum()); ntot += len(ra[1]["flags"])
    assert ntot > S * 100 and (lpf_kind == "reference_table" or ndet > ntot // 2)
    # a front end serves one of the two call styles
    fe = RxFrontEnd(ctx, S, lpf, max_chunks=2)
    fe.push_chunk(d_iq[:, :OUTCHUNK])
    o = bufs(S * 8)
    with pytest.raises(pkg.TrxSigError, match="push / pop"):
        fe.push_detect_demod(d_iq[:, OUTCHUNK:2 * OUTCHUNK], tsc, o["flags"], o["amp"], o["toa"], o["soft"])


def test_fused_front_end_argument_checks(golden):
    """The fused call refuses what it cannot do instead of computing something else: a context at one sample per symbol,
    more than 148 soft bits, more chunks than the object was sized for."""
    import torch
    pkg = _pkg.load()
    from openbts_ttsou_amd import synth
    from openbts_ttsou_amd.frontend import RxFrontEnd, OUTCHUNK
    lpf = synth.design_lpf(961, 260)
    iq = torch.zeros(2, 3 * OUTCHUNK, 2, dtype=torch.int16, device="cuda")
    n = 2 * 16
    o = dict(flags=torch.zeros(n, dtype=torch.uint8, device="cuda"), amp=torch.zeros(n, 2, device="cuda"), toa=torch.zeros(n, device="cuda"),
             soft=torch.zeros(n, 157, device="cuda"))
    ctx1 = pkg.TrxSig(1, 0); ctx1.use_torch_stream()
    fe1 = RxFrontEnd(ctx1, 2, synth.design_lpf(961, 65), max_chunks=3)
    with pytest.raises(pkg.TrxSigError, match="sps == 4"):
        fe1.push_detect_demod(iq, 0, o["flags"], o["amp"], o["toa"], o["soft"])
    ctx4 = pkg.TrxSig(4, 0); ctx4.use_torch_stream()
    fe4 = RxFrontEnd(ctx4, 2, lpf, max_chunks=2)
    with pytest.raises(pkg.TrxSigError, match="bad argument"):
        fe4.push_detect_demod(iq, 0, o["flags"], o["amp"], o["toa"], o["soft"])          # 3 chunks > max_chunks
    with pytest.raises(pkg.TrxSigError, match="bad argument"):
        fe4.push_detect_demod(iq[:, :OUTCHUNK], 0, o["flags"], o["amp"], o["toa"], o["soft"], nsoft=156, soft_stride=157)
    # silence in, nothing detected out, and the object keeps working
    nb, tn = fe4.push_detect_demod(iq[:, :2 * OUTCHUNK], 0, o["flags"], o["amp"], o["toa"], o["soft"], nsoft=148, soft_stride=157)
    torch.cuda.synchronize()
    assert nb == 7 and list(tn) == [0, 1, 2, 3, 4, 5, 6] and not (o["flags"][:2 * nb] & pkg.F_DETECT).any()


@pytest.mark.gpu
def test_objects_keep_their_context_alive():
    """trxsig_destroy on a context that a front end, a back end or a group still lives on takes effect when the last of them is
    gone (the Python wrappers' destructors run in any order): destroying in the 'wrong' order must not touch freed memory, and the
    library keeps working afterwards (a stale hipSetDevice error from a freed context once poisoned an unrelated test)."""
    import torch
    pkg = _pkg.load()
    from openbts_ttsou_amd.frontend import RxFrontEnd, TxBackEnd
    lpf = synth.design_lpf(961, 260)
    ctx = pkg.TrxSig(4, 0); ctx.use_torch_stream()
    fe = RxFrontEnd(ctx, 2, lpf, max_chunks=2)
    be = TxBackEnd(ctx, 2, synth.design_lpf(651, 96), max_bursts=8)
    g = pkg.TrxGroup(ctx, 2, tsc_leg=pkg.TSCLEG_DEMOD)
    ctx.close()                                            # first the context ...
    iq = torch.zeros(2, 864, 2, dtype=torch.int16, device="cuda:0")
    fe.push_chunk(iq)                                      # ... which is still there for its objects
    assert fe.pop_bursts() is not None
    g.close(); be.close(); fe.close()                      # the last one to go takes the context with it
    ctx2 = pkg.TrxSig(4, 0); ctx2.use_torch_stream()
    fe2 = RxFrontEnd(ctx2, 1, lpf, max_chunks=1)
    fe2.push_chunk(iq[:1]); torch.cuda.synchronize()
    fe2.close(); ctx2.close()


def test_refused_front_end_leaves_the_context_unreferenced():
    """trxsig_rxfe_create takes a reference on the context; every refusal path gives it back (round 4: the 2^31-sample refusal
    did not, and that context was never destroyed).  trxsig_live_children counts the references."""
    import ctypes as C
    pkg = _pkg.load()
    ctx = pkg.TrxSig(4, 0)
    L = ctx.L
    L.trxsig_live_children.argtypes = [C.c_void_p]
    assert L.trxsig_live_children(ctx.h) == 0
    lpf = np.ones(64, np.float32)
    h = C.c_void_p()
    L.trxsig_rxfe_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    # 65,535 streams x 65,535 chunks of 2,340 samples: beyond 2^31 samples (refused before anything is allocated)
    rc = L.trxsig_rxfe_create(C.byref(h), ctx.h, 65535, 65535, lpf.ctypes.data, lpf.size, 1, 0)
    assert rc == -1 and not h.value and b"2^31" in L.trxsig_last_error(ctx.h)
    assert L.trxsig_live_children(ctx.h) == 0
    rc = L.trxsig_rxfe_create(C.byref(h), ctx.h, 0, 4, lpf.ctypes.data, lpf.size, 1, 0)      # bad argument: no reference taken
    assert rc == -1 and L.trxsig_live_children(ctx.h) == 0
    rc = L.trxsig_rxfe_create(C.byref(h), ctx.h, 2, 4, lpf.ctypes.data, lpf.size, 1, 0)
    assert rc == 0 and L.trxsig_live_children(ctx.h) == 1
    L.trxsig_rxfe_destroy.argtypes = [C.c_void_p]
    L.trxsig_rxfe_destroy(h)
    assert L.trxsig_live_children(ctx.h) == 0
    ctx.close()
