"""GPU parity of the L1 FEC soft decode (k_fec_viterbi) through the C-ABI: golden vectors
captured from the real reference BitVector / ViterbiR2O4 / Parity code (incl. BitVectorTest.cpp's input),
the CPU oracle on random batches (ties, unknowns, garbage), the UDP-hop quantisation, and the chain
burst -> detect/demod -> FEC -> L2 frame end to end.  Bit-exact."""
import numpy as np
import pytest

import _pkg
import fecbind
import oraclebind
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _pkg.load()


@pytest.fixture(scope="module")
def t(pkg):
    c = pkg.TrxSig(4, 0)
    c.use_torch_stream()
    return c


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def gpu_viterbi(t, soft2d):
    import torch
    nb, n = soft2d.shape
    bits = torch.full((nb, n // 2), 9, dtype=torch.uint8, device="cuda")
    t.fec_viterbi(dev(soft2d.astype(np.float32)), n, nb, bits)
    torch.cuda.synchronize()
    return bits.cpu().numpy()


def gpu_xcch(t, soft, wire):
    import torch
    nb = soft.shape[0] // 4
    frames = torch.full((nb, 23), 9, dtype=torch.uint8, device="cuda"); ok = torch.full((nb,), 9, dtype=torch.uint8, device="cuda")
    t.fec_xcch_decode(dev(soft.astype(np.float32)), nb, frames, ok, wire=wire)
    torch.cuda.synchronize()
    return frames.cpu().numpy(), ok.cpu().numpy()


def gpu_rach(t, soft, wire):
    import torch
    n = soft.shape[0]
    o = [torch.full((n,), 99, dtype=torch.uint8, device="cuda") for _ in range(3)]
    t.fec_rach_decode(dev(soft.astype(np.float32)), n, o[0], o[1], o[2], wire=wire)
    torch.cuda.synchronize()
    return np.stack([x.cpu().numpy() for x in o], axis=1)


def bursts_from_ebits(rng, e4x114):
    """[nb,4,114] e-bits -> [4*nb,148] bursts with the e-bits at 3..59 / 88..144 and junk elsewhere."""
    nb = e4x114.shape[0]
    s = rng.random((4 * nb, 148)).astype(np.float32)
    x = e4x114.reshape(4 * nb, 114)
    s[:, 3:60] = x[:, :57]; s[:, 88:145] = x[:, 57:]
    return s


def test_golden_viterbi(t, golden):
    g = golden("fec.npz")
    got = gpu_viterbi(t, g["kat_c"].astype(np.float32)[None, :])
    assert np.array_equal(got[0], g["kat_u"])                       # CommonLibs/BitVectorTest.cpp:72
    for n in (2, 4, 36, 50, 100, 378, 456):
        assert np.array_equal(gpu_viterbi(t, g["vit%d_in" % n][None, :])[0], g["vit%d_out" % n]), n


def test_golden_xcch_and_rach(t, golden):
    g = golden("fec.npz")
    rng = np.random.default_rng(11)
    frames, ok = gpu_xcch(t, bursts_from_ebits(rng, g["xcch_soft"]), wire=False)
    assert np.array_equal(ok.astype(bool), g["xcch_ok"])
    assert np.array_equal(np.unpackbits(frames, axis=1), g["xcch_dout"])           # d[] even for bad frames
    good = np.flatnonzero(ok)
    assert np.array_equal(np.unpackbits(frames[good], axis=1), g["xcch_d"][good])  # = the frames that were sent
    rs = rng.random((len(g["rach_e"]), 148)).astype(np.float32)
    rs[:, 49:85] = g["rach_e"]
    out = gpu_rach(t, rs, wire=False)
    assert np.array_equal(out[:, 0].astype(bool), g["rach_tail_ok"])
    assert np.array_equal(out[:, 1], g["rach_bsic_out"]) and np.array_equal(out[:, 2], g["rach_ra_out"])


@pytest.mark.parametrize("wire", [False, True])
def test_random_vs_oracle(t, wire):
    o = fecbind.FecOracle()
    rng = np.random.default_rng(123 + wire)
    nb = 1027                                                       # ragged: not a multiple of 4 blocks per wave
    d = rng.integers(0, 2, (nb, 184)).astype(np.uint8)
    hard = np.zeros((nb, 4, 114), np.float32)
    for i in range(nb):
        dd = o.lsb8msb(d[i])
        u = np.zeros(228, np.uint8); u[:184] = dd
        par = (~o.parity(fecbind.XCCH_POLY, 40, dd)) & ((1 << 40) - 1)          # writeParityWord, inverted
        u[184:224] = [(par >> (39 - k)) & 1 for k in range(40)]
        c = o.encode(u)
        k = np.arange(456)
        hard[i, k % 4, 2 * ((49 * k) % 57) + ((k % 8) // 4)] = c
    sig = np.array([0.0, 0.1, 0.2, 0.3, 0.4, 0.6, 1.5])[np.arange(nb) % 7]
    soft = np.clip(hard * 0.8 + 0.1 + rng.normal(0, 1, hard.shape) * sig[:, None, None], 0, 1).astype(np.float32)
    soft[3] = 0.5; soft[4, 1] = 0.5; soft[5] = rng.integers(0, 2, (4, 114))         # ties, a lost burst, hard garbage
    b = bursts_from_ebits(rng, soft)
    frames, ok = gpu_xcch(t, b, wire)
    of, ook = o.xcch_decode_batch(b, wire=wire, nthreads=8)
    assert np.array_equal(ok, ook) and np.array_equal(frames, of)
    assert 0.3 * nb < ok.sum() < 0.8 * nb
    clean = np.flatnonzero(sig <= 0.1); clean = clean[clean > 5]
    assert ok[clean].all() and np.array_equal(np.unpackbits(frames[clean], axis=1), d[clean])
    rs = rng.random((4099, 148)).astype(np.float32)
    assert np.array_equal(gpu_rach(t, rs, wire), o.rach_decode_batch(rs, wire=wire, nthreads=8))
    for n in (2, 38, 456, 1024):
        sv = rng.random((37, n)).astype(np.float32)
        sv[rng.random(sv.shape) < 0.2] = 0.5
        want = np.stack([o.viterbi_decode(sv[i], n // 2) for i in range(len(sv))])
        assert np.array_equal(gpu_viterbi(t, sv), want), n


def test_end_to_end_burst_to_l2_frame(pkg, t):
    """Four normal bursts carrying one XCCH block -> modulate -> channel -> detect + demodulate -> FEC: the
    L2 frame comes back; the GPU chain equals the oracle chain bit for bit (wire quantisation on)."""
    import torch
    sps, tsc, nblk = 4, 5, 64
    o = fecbind.FecOracle(); so = oraclebind.Oracle(sps)
    rng = np.random.default_rng(2026)
    d = rng.integers(0, 2, (nblk, 184)).astype(np.uint8)
    bits = synth.normal_bits(rng, 4 * nblk, tsc)
    for i in range(nblk):
        dd = o.lsb8msb(d[i]); u = np.zeros(228, np.uint8); u[:184] = dd
        par = (~o.parity(fecbind.XCCH_POLY, 40, dd)) & ((1 << 40) - 1)
        u[184:224] = [(par >> (39 - k)) & 1 for k in range(40)]
        c = o.encode(u)
        k = np.arange(456)
        e = np.zeros((4, 114), np.uint8); e[k % 4, 2 * ((49 * k) % 57) + ((k % 8) // 4)] = c
        bits[4 * i:4 * i + 4, 3:60] = e[:, :57]; bits[4 * i:4 * i + 4, 88:145] = e[:, 57:]
        bits[4 * i:4 * i + 4, 60] = 1; bits[4 * i:4 * i + 4, 87] = 1            # stealing flags (fec:716-717)
    B = 4 * nblk
    x, off, length, meta = synth.bursts_from_bits(bits, sps, seed=5, sigmas=(0.0, 0.1, 0.25))
    from util import GpuBatch
    gb = GpuBatch(x, off, length, nsoft=148, stride=148)
    t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, nsoft=148, soft_stride=148)
    frames = torch.zeros(nblk, 23, dtype=torch.uint8, device="cuda"); ok = torch.zeros(nblk, dtype=torch.uint8, device="cuda")
    t.fec_xcch_decode(gb.soft, nblk, frames, ok, wire=True)
    torch.cuda.synchronize()
    frames, ok = frames.cpu().numpy(), ok.cpu().numpy()
    _, _, _, soft = so.normal_batch(x, off, length, tsc, nsoft=148, nthreads=8)
    of, ook = o.xcch_decode_batch(soft[:, :148], wire=True, nthreads=4)
    assert np.array_equal(ok, ook) and np.array_equal(frames, of)
    assert ok.sum() >= 0.9 * nblk
    good = np.flatnonzero(ok)
    assert np.array_equal(np.unpackbits(frames[good], axis=1), d[good])


def test_tch_facch(t, golden):
    """TCH/FACCH full rate: 8-burst diagonal deinterleaver, class-1 Viterbi, class-2 slicing, 3-bit parity +
    tail, stealing flag, and the FACCH (XCCH) decode of the same blocks -- golden vectors of the reference's
    decodeTCH steps, then random bursts against the oracle with and without the UDP-hop quantisation."""
    import torch
    from test_fec_oracle import tch_bursts
    g = golden("fec.npz")
    o = fecbind.FecOracle()
    rng = np.random.default_rng(44)

    def gpu(b, wire):
        nbl = b.shape[0] // 4 - 1
        tch = torch.full((nbl, 33), 7, dtype=torch.uint8, device="cuda")
        outs = [torch.full((nbl,), 7, dtype=torch.uint8, device="cuda") for _ in range(3)]
        facch = torch.full((nbl, 23), 7, dtype=torch.uint8, device="cuda")
        t.fec_tch_decode(dev(b.astype(np.float32)), b.shape[0], tch, outs[0], outs[1], facch=facch, facch_ok=outs[2], wire=wire)
        torch.cuda.synchronize()
        return dict(tch=tch.cpu().numpy(), good=outs[0].cpu().numpy(), stolen=outs[1].cpu().numpy(),
                    facch=facch.cpu().numpy(), facch_ok=outs[2].cpu().numpy())

    b = tch_bursts(rng, g["tch_soft"])
    r = gpu(b, False)
    assert np.array_equal(r["good"].astype(bool), g["tch_good"])
    assert np.array_equal(np.unpackbits(r["tch"], axis=1)[:, :260], g["tch_dout"])
    for wire in (False, True):
        b = rng.random((4 * 1030 + 3, 148)).astype(np.float32)                 # ragged tail: the last 3 bursts are unused
        b[rng.random(b.shape) < 0.1] = 0.5
        # a third of the blocks carry valid TCH frames
        nblk = b.shape[0] // 4 - 1
        c = np.zeros((nblk, 456), np.float32)
        for m in range(nblk):
            d = rng.integers(0, 2, 260).astype(np.uint8)
            u = np.zeros(189, np.uint8)
            u[:91] = d[0:182:2]; u[184:93:-1] = d[1:182:2]
            par = (~o.parity(0x0b, 3, d[:50])) & 7
            u[91:94] = [(par >> 2) & 1, (par >> 1) & 1, par & 1]
            c[m, :378] = o.encode(u); c[m, 378:] = d[182:]
        valid = np.arange(nblk) % 3 == 0
        bb = tch_bursts(rng, np.clip(c * 0.8 + 0.1 + rng.normal(0, 0.15, c.shape), 0, 1))
        mask = np.repeat(valid, 4)
        # keep the random junk where the block is not valid (both halves of the diagonal)
        for m in np.flatnonzero(valid):
            b[4 * m:4 * m + 8] = np.where(np.isin(np.arange(148), np.r_[3:60, 88:145])[None, :], bb[4 * m:4 * m + 8], b[4 * m:4 * m + 8])
        got = gpu(b, wire)
        want = o.tch_decode_batch(b, wire=wire, nthreads=8)
        for k in ("tch", "good", "stolen", "facch", "facch_ok"):
            assert np.array_equal(got[k], want[k]), (k, wire)
        assert got["good"][valid].mean() > 0.5


def test_argument_checks_and_empty_batches(pkg, t):
    """Bad arguments are rejected with TRXSIG_EINVAL (no launch); empty batches are no-ops."""
    import torch
    L = pkg.lib()
    soft = torch.rand(8, 148, device="cuda")
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    p = lambda x: x.data_ptr()
    assert L.trxsig_fec_xcch_decode_batch(t.h, p(soft), 148, 0, 1, p(out), p(out)) == 0
    assert L.trxsig_fec_xcch_decode_batch(t.h, p(soft), 100, 2, 1, p(out), p(out)) != 0          # stride < 148
    assert L.trxsig_fec_xcch_decode_batch(t.h, None, 148, 2, 1, p(out), p(out)) != 0
    assert L.trxsig_fec_rach_decode_batch(t.h, p(soft), 148, 0, 1, p(out), p(out), p(out)) == 0
    assert L.trxsig_fec_rach_decode_batch(t.h, p(soft), 148, 8, 1, None, p(out), p(out)) != 0
    assert L.trxsig_fec_tch_decode_batch(t.h, p(soft), 148, 7, 1, p(out), p(out), None, None, p(out)) == 0   # < 8 bursts: no block
    assert L.trxsig_fec_tch_decode_batch(t.h, p(soft), 148, 8, 1, p(out), p(out), p(out), None, p(out)) != 0 # facch without its flag
    assert L.trxsig_fec_viterbi_batch(t.h, p(soft), 37, 148, 2, p(out), 32) != 0                  # odd length
    assert L.trxsig_fec_viterbi_batch(t.h, p(soft), 148, 148, 0, p(out), 74) == 0
    assert b"bad argument" in L.trxsig_last_error(t.h)
    torch.cuda.synchronize()


def test_end_to_end_access_burst_to_ra(pkg, t):
    """Access bursts carrying RA + BSIC-masked parity -> modulate -> channel with unknown delay -> detectRACHBurst +
    demodulateBurst -> RACH FEC: RA and BSIC come back; GPU chain == oracle chain bit for bit."""
    import torch
    sps, n = 4, 256
    o = fecbind.FecOracle(); so = oraclebind.Oracle(sps)
    rng = np.random.default_rng(77)
    ra = rng.integers(0, 256, n); bsic = rng.integers(0, 64, n)
    bits = synth.rach_bits(rng, n)
    for i in range(n):
        u = np.zeros(18, np.uint8)
        u[:8] = o.lsb8msb(np.array([(ra[i] >> (7 - k)) & 1 for k in range(8)], np.uint8))
        sent = (~(o.parity(fecbind.RACH_POLY, 6, u[:8]) ^ int(bsic[i]))) & 0x3f
        u[8:14] = [(sent >> (5 - k)) & 1 for k in range(6)]
        bits[i, 49:85] = o.encode(u)
    x, off, length, meta = synth.bursts_from_bits(bits, sps, seed=6, sigmas=(0.0, 0.1, 0.2), max_delay=1.5)
    from util import GpuBatch
    gb = GpuBatch(x, off, length, nsoft=148, stride=148)
    t.detect_demod_rach(gb.x, gb.off, gb.len, gb.flags, gb.amp, gb.toa, gb.soft, energy_thresh=-1.0, nsoft=148, soft_stride=148)
    out = [torch.zeros(n, dtype=torch.uint8, device="cuda") for _ in range(3)]
    t.fec_rach_decode(gb.soft, n, out[0], out[1], out[2], wire=True)
    torch.cuda.synchronize()
    got = np.stack([v.cpu().numpy() for v in out], axis=1)
    ok, amp, toa, soft = so.rach_batch(x, off, length, nsoft=148, nthreads=8)
    assert np.array_equal(got, o.rach_decode_batch(soft[:, :148], wire=True, nthreads=4))
    det = (gb.flags.cpu().numpy() & pkg.F_DETECT) != 0
    good = det & (got[:, 0] == 1) & (got[:, 1] == bsic)
    assert good.sum() >= 0.85 * n and np.array_equal(got[good, 2], ra[good])


TSCS = ["00100101110000100010010111", "00101101110111100010110111", "01000011101110100100001110", "01000111101101000100011110",
        "00011010111001000001101011", "01001110101100000100111010", "10100111110110001010011111", "11101111000100101110111100"]


def gpu_xcch_encode(t, frames, tsc):
    import torch
    nb = len(frames)
    bits = torch.full((4 * nb, 148), 7, dtype=torch.uint8, device="cuda")
    t.fec_xcch_encode(dev(frames), nb, tsc, bits)
    torch.cuda.synchronize()
    return bits.cpu().numpy()


def test_xcch_encode(pkg, t, golden):
    """XCCH L1 encode (k_fec_xcch_encode): the reference's own e-bits (golden), the oracle on random frames
    for every TSC, ragged batch sizes, argument checks."""
    g = golden("fec.npz")
    frames = np.packbits(g["xcch_d"], axis=1)
    b = gpu_xcch_encode(t, frames, 5).reshape(-1, 4, 148)
    e = np.concatenate([b[:, :, 3:60], b[:, :, 88:145]], axis=2)
    assert np.array_equal(e, g["xcch_hard"])
    o = fecbind.FecOracle(); rng = np.random.default_rng(77)
    for tsc, nb in enumerate([1, 2, 3, 63, 64, 65, 257, 1000]):
        fr = rng.integers(0, 256, (nb, 23)).astype(np.uint8)
        fr[0] = 0; fr[-1] = 255
        tb = np.array([int(c) for c in TSCS[tsc]], np.uint8)
        want = np.concatenate([o.xcch_encode(f, tb) for f in fr])
        assert np.array_equal(gpu_xcch_encode(t, fr, tsc), want), (tsc, nb)
    import torch
    z = torch.zeros(4, 148, dtype=torch.uint8, device="cuda")
    t.fec_xcch_encode(None, 0, 0, None)
    for bad in (-1, 8):
        with pytest.raises(pkg.TrxSigError):
            t.fec_xcch_encode(z, 1, bad, z)
    with pytest.raises(pkg.TrxSigError):
        t.fec_xcch_encode(None, 1, 0, z)


def test_closed_loop_l2_to_l2(pkg, t):
    """L2 frames -> XCCH encode -> GMSK modulate -> noise -> TSC detect + demodulate -> XCCH decode, all on the
    card through the C-ABI: every frame comes back and the parity check passes."""
    import torch
    sps, tsc, nblk = 4, 2, 512
    B = 4 * nblk
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    frames = torch.randint(0, 256, (nblk, 23), dtype=torch.uint8, device="cuda", generator=g)
    bits = torch.zeros(B, 148, dtype=torch.uint8, device="cuda")
    t.fec_xcch_encode(frames, nblk, tsc, bits)
    guard = torch.full((B,), 8, dtype=torch.int32, device="cuda")
    n = sps * 156
    off = (torch.arange(B, dtype=torch.int32, device="cuda") * n).contiguous()
    length = torch.full((B,), n, dtype=torch.int32, device="cuda")
    x = torch.zeros(B * n, 2, dtype=torch.float32, device="cuda")
    t.modulate(bits, guard, x, off)
    x += 0.15 * torch.randn(x.shape, device="cuda", generator=g)
    flags = torch.zeros(B, dtype=torch.uint8, device="cuda"); amp = torch.zeros(B, 2, device="cuda"); toa = torch.zeros(B, device="cuda")
    soft = torch.zeros(B, 148, device="cuda")
    t.detect_demod_normal(x, off, length, tsc, flags, amp, toa, soft, nsoft=148, soft_stride=148)
    out = torch.zeros_like(frames); ok = torch.zeros(nblk, dtype=torch.uint8, device="cuda")
    t.fec_xcch_decode(soft, nblk, out, ok, wire=True)
    torch.cuda.synchronize()
    assert bool(((flags & pkg.F_DETECT) != 0).all()) and bool(ok.all())
    assert torch.equal(out, frames)
