"""The TRANSMIT half of the Transceiver group (include/trxsig_trxgroup.h, csrc/trxsig_grouptx.hip): addRadioVector /
pushRadioVector (Transceiver/Transceiver.cpp:100-113, 138-181) for S ARFCNs with the priority queue, the stale-burst dump and
the filler table [FN % modulus][TN] on the device.  Checked against
  (1) S independent one-ARFCN objects (include/trxsig_transceiver.h: the reference's own container, std::priority_queue), fed the
      same 154-byte datagrams in the same order, one burst per call;
  (2) oracle/transceiver_model.py's add_radio_vector / push_radio_vector on the CPU oracle;
on every (timeslot, ARFCN) cell: came-from-the-queue flag and the modulated, power-scaled burst that reaches the transmit FIFO,
value for value -- and the int16 stream the fused transmit back end makes of the group's output against the oracle chain
(modulateBurst -> scaleVector -> polyphaseResampleVector 96 : 65 -> x 13500 -> int16) on the model's bursts.
S = 128 ARFCNs x 208 frames, channel combinations I / II / IV / V / VI / VII / NONE (filler moduli 26 / 51 / 102), bursts on
time, LATE (stale on arrival: they must still land in the filler table), EARLY (many frames ahead), DUPLICATES (two bursts
for one timestamp: the heap's shape decides which goes out), RSSI over the whole signed byte, pushes of 1 ... 40 timeslots
starting on any timeslot, a start just below the hyperframe wrap.  A third case keeps ~120 bursts queued per ARFCN (ties among
them): every level of the LDS heap moves (csrc/trxsig_txq_lds.h: tx_heap_push / tx_heap_pop) is walked.  A fourth queues bursts a
third of a hyperframe from the rest: the kernels' slow path (the queue's arrays in memory, trxsig_txq.h's moves)."""
import os

import numpy as np
import pytest

import _pkg
import oraclebind
import synth
import transceiver_model as tm
from test_gpu_trxgroup import configure

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _pkg.load()


def datagram(tn, fn, rssi, bits):
    d = np.zeros(154, np.uint8)
    d[0] = tn
    d[1:5] = [(fn >> 24) & 255, (fn >> 16) & 255, (fn >> 8) & 255, fn & 255]
    d[5] = np.uint8(rssi & 255)
    d[6:] = bits
    return d


def traffic(rng, S, fn, chan_used, tame=(), deep=False):
    """The datagrams the GSM core sends while frame `fn` is on the air: (arfcn, datagram) in arrival order.  RSSI (an
    attenuation in dB) covers the whole signed byte except on the ARFCNs in `tame`, whose int16 stream is compared: a negative
    RSSI is a gain of up to 10^12, the samples leave the int16 range and the reference's (short) cast is undefined there."""
    out = []
    H = tm.HYPERFRAME
    for a in range(S):
        for tn in range(8):
            if not chan_used[a][tn]:
                continue
            r = rng.random()
            rssi = int(rng.integers(-128, 128)) if (rng.random() < 0.3 and a not in tame) else int(rng.integers(0, 40))
            bits = lambda: rng.integers(0, 2, 148).astype(np.uint8) * (1 + 2 * int(rng.integers(0, 2)))   # (1 or 3: only bit 0 counts)
            if r < 0.55:
                out.append((a, datagram(tn, (fn + 3) % H, rssi, bits())))
            elif r < 0.62:                                               # a duplicate: two bursts for one timestamp (and a third now and then)
                for _ in range(2 + (rng.random() < 0.3)):
                    out.append((a, datagram(tn, (fn + 3) % H, rssi, bits())))
            elif r < 0.68:                                               # late: its slot has gone by
                out.append((a, datagram(tn, (fn - 1 - int(rng.integers(0, 4))) % H, rssi, bits())))
            elif r < 0.73:                                               # early
                out.append((a, datagram(tn, (fn + 4 + int(rng.integers(0, 25))) % H, rssi, bits())))
            if deep and rng.random() < 0.9:                              # a DEEP queue: ~120 bursts waiting per ARFCN (seven heap levels),
                for _ in range(1 + (rng.random() < 0.15)):               # ties among them
                    out.append((a, datagram(tn, (fn + 4 + int(rng.integers(0, 18))) % H, rssi, bits())))
    order = rng.permutation(len(out))
    return [out[i] for i in order]


@pytest.mark.parametrize("sps,S,frames,fn0,deep", [(1, 128, 208, tm.HYPERFRAME - 90, False), (4, 8, 40, 1234, False),
                                                   (1, 20, 90, tm.HYPERFRAME - 50, True), (1, 20, 30, 4321, "far")])
def test_group_transmit_half(pkg, golden, sps, S, frames, fn0, deep):
    import torch
    from openbts_ttsou_amd.frontend import TxBackEnd, OUTHISTORY
    dev = torch.device("cuda:0")
    o = oraclebind.Oracle(sps)
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    grp = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD)        # group A: what goes out, cell by cell
    grp_b = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD)      # group B: the same straight into the transmit back end
    lpf = golden("resample.npz")["lpf651_gain96"]
    be = TxBackEnd(ctx, S, lpf, max_bursts=64)
    objs = [pkg.TrxHost(sps, 0) for _ in range(S)]
    models = [tm.TransceiverModel(o) for _ in range(S)]
    chan_used = []
    for a in range(S):
        ra = configure(lambda m: grp.control(a, m), a)
        assert ra == configure(lambda m: grp_b.control(a, m), a) == configure(objs[a].control, a) == configure(models[a].control, a)
        chan_used.append([models[a].chan_type[tn] != tm.NONE or (a + tn) % 5 == 0 for tn in range(8)])   # (some traffic on idle slots too)
    rng = np.random.default_rng(2024 + sps + 1000 * int(os.environ.get("TRXSIG_TX_SOAK_SEED", "0")))   # (tools/r05_tx_soak.sh: more seeds)
    H = tm.HYPERFRAME
    n_slots_total = frames * 8
    pos = first = int(rng.integers(0, 8))                       # the first push starts on any timeslot
    fed_until = -1                                              # last frame whose traffic has been added
    streams = (0, 1, S - 1)                                     # int16 stream checked for these ARFCNs
    hist = {s: np.zeros(2 * 65 * sps, np.complex64) for s in streams}
    send = {s: np.zeros(0, np.complex64) for s in streams}
    inchunk = 65 * 9 * sps
    n_cells = n_fq = n_int16 = 0
    while pos < n_slots_total:
        n = int(rng.integers(1, 41))
        n = min(n, n_slots_total - pos)
        last_frame = (pos + n - 1) // 8
        # the core's traffic up to the frame the push ends in (adds before pushes, as the two service loops interleave)
        dgs, arf = [], []
        for f in range(fed_until + 1, last_frame + 1):
            for a, d in traffic(rng, S, (fn0 + f) % H, chan_used, tame=streams, deep=deep is True):
                dgs.append(d); arf.append(a)
            if deep == "far" and f in (2, 11):
                # a burst a third of a hyperframe away, then one of its timestamp's neighbours: the kernels' packed queue entries
                # cannot say such a time (csrc/trxsig_txq_lds.h) -- frame 2: the CALL takes the slow path (the host sees the
                # datagram), afterwards ARFCN 3's workgroup does whenever it loads its queue (the entry stays queued); frame 11:
                # a call whose first datagram is the far one (every other datagram is far from the reference then)
                at = len(dgs) if f == 2 else 0
                dgs.insert(at, datagram(5, (fn0 + f + H // 3) % H, 7, rng.integers(0, 2, 148).astype(np.uint8))); arf.insert(at, 3)
        fed_until = max(fed_until, last_frame)
        if dgs:
            dg = np.stack(dgs); ar = np.array(arf, np.int32)
            grp.add_bursts(dg, ar); grp_b.add_bursts(dg, ar)
            for a, d in zip(arf, dgs):
                tn, fn, rssi, bits = objs[a].decode_tx_datagram(d.tobytes())
                assert (tn, fn) == (int(d[0]), int.from_bytes(d[1:5].tobytes(), "big"))
                objs[a].add_radio_vector(bits, rssi, tn, fn)
                mt, mf, mr, mb = models[a].decode_tx_datagram(d.tobytes())
                models[a].add_radio_vector(mb, mr, mt, mf)
        fn, tn = (fn0 + pos // 8) % H, pos % 8
        bits_d, gain_d, fq_d = grp.push(fn, tn, n)
        grp_b.push_txbe(be, fn, tn, n)
        got_iq = be.pop_samples()
        # the group's output, modulated on the device exactly as addRadioVector does (modulateBurst + scaleVector)
        guard = np.array([8 + (((tn + t) % 8) % 4 == 0) for t in range(n)], np.int32)
        guard_all = torch.from_numpy(np.tile(guard, S)).to(dev)
        length = (sps * (148 + np.tile(guard, S))).astype(np.int32)
        off = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.int32)
        out = torch.zeros(int(length.sum()), 2, device=dev)
        ctx.modulate(bits_d.reshape(S * n, 148).contiguous(), guard_all, out, torch.from_numpy(off).to(dev), gain=gain_d.reshape(-1).contiguous())
        xs = out.cpu().numpy().view(np.complex64).ravel()
        fq = fq_d.cpu().numpy()
        for t in range(n):
            sfn, stn = (fn0 + (pos + t) // 8) % H, (pos + t) % 8
            for a in range(S):
                want, wfq = objs[a].push_radio_vector(stn, sfn)
                mwant, mfq = models[a].push_radio_vector(stn, sfn)
                k = a * n + t
                mine = xs[off[k]:off[k] + length[k]]
                assert bool(fq[a, t]) == wfq == mfq, (a, sfn, stn)
                assert np.array_equal(mine, want) and np.array_equal(mine, mwant), (a, sfn, stn, grp.tx_queue_size(a), len(models[a].queue))
                n_fq += wfq
                if a in send:
                    send[a] = np.concatenate([send[a], mwant])
        n_cells += n * S
        # the int16 stream of group B through the fused back end against the oracle chain on the model's bursts
        nch = len(send[streams[0]]) // inchunk
        if nch == 0:
            assert got_iq is None
        else:
            iq = got_iq.cpu().numpy()
            for s in streams:
                tr = send[s][:nch * inchunk]
                y = o.polyphase_resample(np.concatenate([hist[s], tr]), 96, 65 * sps, lpf)
                y = o.scale_vector(y, complex(13500.0, 0.0))
                want = np.stack([np.trunc(y.real), np.trunc(y.imag)], axis=1).astype(np.int16)[OUTHISTORY:]
                assert iq[s].shape == want.shape and np.array_equal(iq[s], want), (s, pos)
                hist[s] = tr[-2 * 65 * sps:]
                send[s] = send[s][nch * inchunk:]
                n_int16 += want.shape[0]
        pos += n
    assert pos == n_slots_total and n_cells == (n_slots_total - first) * S
    assert n_fq > n_cells // 4 and n_int16 > 1000
    deepest = 0
    for a in (0, S // 2, S - 1):                                # what is left queued (the early bursts) agrees too, nothing was dropped
        q, dropped = grp.tx_queue_size(a)
        assert q == objs[a].L.trxsig_trx_queue_size(objs[a].h) == len(models[a].queue) and not dropped
        deepest = max(deepest, q)
    assert deep is not True or 64 < deepest < 256, deepest
    for x in objs:
        x.close()
    be.close(); grp.close(); grp_b.close(); ctx.close()


def test_group_transmit_refusals_and_overflow(pkg):
    """Bad datagrams refuse the whole call; a queue that fills up drops the overflow and says so."""
    import torch
    ctx = pkg.TrxSig(1, 0); ctx.use_torch_stream()
    grp = pkg.TrxGroup(ctx, 4, tsc_leg=pkg.TSCLEG_DEMOD)
    bits = np.ones(148, np.uint8)
    ok = datagram(3, 100, 0, bits)
    for bad, arf in ((datagram(8, 100, 0, bits), 0), (datagram(3, tm.HYPERFRAME, 0, bits), 0), (ok, 4), (ok, -1)):
        with pytest.raises(pkg.TrxSigError):
            grp.add_bursts(np.stack([ok, bad]), np.array([0, arf], np.int32))
    assert grp.tx_queue_size(0) == (0, False)
    grp.add_bursts(np.stack([datagram(k % 8, 1000 + k // 8, 0, bits) for k in range(300)]), np.full(300, 2, np.int32))
    assert grp.tx_queue_size(2) == (256, True) and grp.tx_queue_size(1) == (0, False)
    b, g, fq = grp.push(1000, 0, 64)                            # the 256 that fitted go out in order, ARFCN 1 sends the dummy burst
    torch.cuda.synchronize()
    fq = fq.cpu().numpy()
    assert fq[2].all() and not fq[1].any()
    dummy = np.array([int(c) for c in tm.DUMMY_BURST], np.uint8)
    assert np.array_equal(b.cpu().numpy()[1, 5], dummy) and float(g.cpu().numpy()[1, 5]) == 1.0
    grp.close(); ctx.close()


def test_push_txbe_refusals_leave_the_queue_alone(pkg):
    """trxsig_trxgroup_push_txbe settles everything the back end could refuse BEFORE it pops the transmit queue: a back end with
    another stream count, a back end on another context, and a back end whose send buffers cannot take the push all answer
    EINVAL with the queued bursts still queued (the caller's deadline clock has not advanced either); the same push into a
    back end with room then sends them."""
    import torch
    from openbts_ttsou_amd.frontend import TxBackEnd
    from openbts_ttsou_amd import synth
    sps, S = 1, 4
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    other = pkg.TrxSig(sps, 0); other.use_torch_stream()
    grp = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD)
    lpf = synth.design_lpf(961, 65 * sps)
    bits = np.ones(148, np.uint8)
    grp.add_bursts(np.stack([datagram(k % 8, 2000 + k // 8, 0, bits) for k in range(16)]), np.full(16, 1, np.int32))
    assert grp.tx_queue_size(1) == (16, False)
    wrong_s = TxBackEnd(ctx, S + 1, lpf, max_bursts=16)
    wrong_ctx = TxBackEnd(other, S, lpf, max_bursts=16)
    small = TxBackEnd(ctx, S, lpf, max_bursts=4)             # 8 slots do not fit a back end made for 4 bursts per push
    good = TxBackEnd(ctx, S, lpf, max_bursts=16)
    for be in (wrong_s, wrong_ctx, small):
        with pytest.raises(pkg.TrxSigError):
            grp.push_txbe(be, 2000, 0, 8)
        assert grp.tx_queue_size(1) == (16, False)
    # a full back end: fill `good` until it refuses, then the group's push is refused too and the queue is untouched
    filler = np.zeros((S, 16, 148), np.uint8)
    guard = np.array([8 + ((t % 8) % 4 == 0) for t in range(16)], np.int32)
    n_fill = 0
    while True:
        try:
            good.push_bursts(filler, guard)
            n_fill += 1
        except pkg.TrxSigError:
            break
        assert n_fill < 64
    with pytest.raises(pkg.TrxSigError):
        grp.push_txbe(good, 2000, 0, 16)
    assert grp.tx_queue_size(1) == (16, False)
    roomy = TxBackEnd(ctx, S, lpf, max_bursts=16)
    grp.push_txbe(roomy, 2000, 0, 16)
    torch.cuda.synchronize()
    assert grp.tx_queue_size(1) == (0, False)
    for be in (wrong_s, wrong_ctx, small, good, roomy):
        be.close()
    grp.close(); ctx.close(); other.close()


def test_staged_add_equals_the_copying_add(pkg):
    """trxsig_trxgroup_tx_staging + _add_staged (the datagrams received straight into the group's pinned block) against
    trxsig_trxgroup_add_bursts on the same datagrams -- more than one round of the ingest kernel (9,900 > 8,192 per round), ARFCNs
    interleaved, late, early and duplicate bursts, a run that overflows one ARFCN's queue: the pushes hand out the same bits, gains
    and from-queue marks, the queues end up the same size, the same ARFCN reports the drop; a bad header in the staged block
    refuses the batch and leaves the block with the caller."""
    import torch
    rng = np.random.default_rng(12)
    S, F = 20, 60                                           # 20 ARFCNs (two workgroups of the ingest kernel, the second partly filled)
    ctx = pkg.TrxSig(1, 0); ctx.use_torch_stream()
    ga = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD)
    gb = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD)
    for g in (ga, gb):
        for a in range(S):
            configure(lambda c, a=a: g.control(a, c), a)
    n = S * 8 * F
    dg = np.zeros((n, 154), np.uint8)
    arf = rng.integers(0, S, n).astype(np.int32)
    fn = 3000 + rng.integers(-3, F + 3, n)                  # some already stale, some beyond the frames pushed below
    dg[:, 0] = rng.integers(0, 8, n)
    dg[:, 1] = fn >> 24; dg[:, 2] = (fn >> 16) & 255; dg[:, 3] = (fn >> 8) & 255; dg[:, 4] = fn & 255
    dg[:, 5] = rng.integers(-128, 128, n).astype(np.int8).view(np.uint8)
    dg[:, 6:] = rng.integers(0, 2, (n, 148))
    extra = np.repeat(dg[:1], 300, axis=0); extra[:, 1:5] = [0, 0, 0x0c, 0x00]       # 300 more for one ARFCN: its queue overflows
    dg = np.concatenate([dg, extra]); arf = np.concatenate([arf, np.full(300, 7, np.int32)])
    n = len(dg)
    assert n > 8192
    ga.add_bursts(dg, arf)
    d, a = gb.tx_staging(n)
    d[:] = dg; a[:] = arf
    bad = d[5].copy(); d[5, 0] = 9                          # TN 9: the staged batch is refused, nothing queued, the block stays ours
    with pytest.raises(pkg.TrxSigError):
        gb.add_staged(n)
    assert gb.tx_queue_size(3) == (0, False)
    d[5] = bad
    gb.add_staged(n)
    for a_ in range(S):
        assert ga.tx_queue_size(a_) == gb.tx_queue_size(a_)
    assert ga.tx_queue_size(7)[1]                           # the overflow was reported
    for f0 in (2990, 3010, 3030):
        ra = ga.push(f0, 0, 8 * 20); rb = gb.push(f0, 0, 8 * 20)
        torch.cuda.synchronize()
        for xa, xb in zip(ra, rb):
            assert torch.equal(xa, xb)
    for a_ in range(S):
        assert ga.tx_queue_size(a_) == gb.tx_queue_size(a_)
    ga.close(); gb.close(); ctx.close()


def test_one_add_of_twenty_frames_then_one_push(pkg):
    """Three rounds of the ingest kernel in ONE add call (128 ARFCNs x 20 frames x 8 timeslots = 20,480 datagrams in a random arrival
    order, every (ARFCN, frame, timeslot) once: 160 entries a queue), then ONE push of the 160 timeslots: every cell must be exactly
    the burst that was sent for it (bits, gain pow(10, -RSSI/10) with the reference's integer division), every one from the queue,
    the queues empty afterwards -- an expectation that needs no second implementation."""
    import torch
    rng = np.random.default_rng(99)
    S, F, fn0 = 128, 20, tm.HYPERFRAME - 7                  # across the hyperframe wrap
    ctx = pkg.TrxSig(1, 0); ctx.use_torch_stream()
    grp = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD)
    for a in range(S):
        configure(lambda c, a=a: grp.control(a, c), a)
    n = S * F * 8
    arf = np.repeat(np.arange(S, dtype=np.int32), F * 8)
    f = np.tile(np.repeat(np.arange(F), 8), S)
    tn = np.tile(np.arange(8), S * F)
    fn = (fn0 + f) % tm.HYPERFRAME
    rssi = rng.integers(0, 60, n)
    dg = np.zeros((n, 154), np.uint8)
    dg[:, 0] = tn
    dg[:, 1] = fn >> 24; dg[:, 2] = (fn >> 16) & 255; dg[:, 3] = (fn >> 8) & 255; dg[:, 4] = fn & 255
    dg[:, 5] = rssi
    dg[:, 6:] = rng.integers(0, 2, (n, 148))
    order = rng.permutation(n)
    grp.add_bursts(dg[order], arf[order])
    bits_d, gain_d, fq_d = grp.push(fn0, 0, F * 8)
    torch.cuda.synchronize()
    bits = bits_d.cpu().numpy().reshape(S, F * 8, 148)
    gain = gain_d.cpu().numpy().reshape(S, F * 8)
    fq = fq_d.cpu().numpy().reshape(S, F * 8)
    assert fq.all()
    assert np.array_equal(bits, dg[:, 6:].reshape(S, F * 8, 148))
    want_gain = np.array([np.float32(10.0 ** int(-int(r) / 10)) for r in rssi], np.float32).reshape(S, F * 8)   # (C's -RSSI / 10 truncates towards zero)
    assert np.array_equal(gain, want_gain)
    for a in (0, 63, 127):
        assert grp.tx_queue_size(a) == (0, False)
    grp.close(); ctx.close()


def test_pending_ingest_taken_into_the_push_or_launched_alone(pkg):
    """An add call's ingest stays pending until the push that follows takes it into its own launch (k_group_tx<true, true>); another
    add, a queue-size query or a push that starts far from the datagrams' frames launch it on its own first.  Same datagrams four
    ways -- add + push (one launch); add + add (the halves) + push; add + tx_queue_size + push; add + a push 70,000 frames earlier
    (too far to share the launch: nothing is due, everything stays queued) + the real push -- must hand out the same bits, gains and
    from-queue marks and leave the same queues."""
    import torch
    rng = np.random.default_rng(41)
    S, F, fn0 = 24, 6, 90000
    ctx = pkg.TrxSig(1, 0); ctx.use_torch_stream()
    groups = [pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD) for _ in range(4)]
    for g in groups:
        for a in range(S):
            configure(lambda c, a=a: g.control(a, c), a)
    n = S * 8 * F
    arf = rng.integers(0, S, n).astype(np.int32)
    fn = fn0 + rng.integers(-2, F + 2, n)                   # some stale, some beyond the frames pushed
    dg = np.zeros((n, 154), np.uint8)
    dg[:, 0] = rng.integers(0, 8, n)
    dg[:, 1] = fn >> 24; dg[:, 2] = (fn >> 16) & 255; dg[:, 3] = (fn >> 8) & 255; dg[:, 4] = fn & 255
    dg[:, 5] = rng.integers(0, 50, n)
    dg[:, 6:] = rng.integers(0, 2, (n, 148))
    ga, gb, gc, gd = groups
    ga.add_bursts(dg, arf)
    gb.add_bursts(dg[:n // 2], arf[:n // 2]); gb.add_bursts(dg[n // 2:], arf[n // 2:])
    gc.add_bursts(dg, arf); sizes = [gc.tx_queue_size(a) for a in range(S)]
    gd.add_bursts(dg, arf)
    bits_e, gain_e, fq_e = gd.push(fn0 - 70000, 3, 5)       # nothing is due 70,000 frames earlier: five filler slots
    torch.cuda.synchronize()
    assert not fq_e.cpu().numpy().any()
    assert [gd.tx_queue_size(a) for a in range(S)] == sizes and sum(q for q, _ in sizes) == n
    outs = []
    for g in groups:
        b, ga_, fq = g.push(fn0, 0, 8 * F)
        torch.cuda.synchronize()
        outs.append((b.cpu().numpy().copy(), ga_.cpu().numpy().copy(), fq.cpu().numpy().copy(), [g.tx_queue_size(a) for a in range(S)]))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2]) and o[3] == outs[0][3]
    assert outs[0][2].sum() > n // 4                       # (random (ARFCN, frame, timeslot) triples collide: a slot sends one of its bursts)
    for g in groups:
        g.close()
    ctx.close()
