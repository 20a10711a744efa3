"""L1 FEC soft decode, CPU side: the restatement oracle/fec_oracle.c against (1) the
golden vectors captured from the real reference BitVector / ViterbiR2O4 / Parity code -- including the
input string of the reference's own CommonLibs/BitVectorTest.cpp:72 -- everywhere, and (2) that
reference itself on random inputs where oracle/_ref/libref_fec.so exists (build container)."""
import numpy as np
import pytest

import fecbind
import reffec


@pytest.fixture(scope="module")
def o():
    return fecbind.FecOracle()


def test_golden_viterbi_and_coder(o, golden):
    g = golden("fec.npz")
    assert np.array_equal(o.viterbi_decode(g["kat_c"].astype(np.float32), len(g["kat_u"])), g["kat_u"])
    for n in (2, 4, 36, 50, 100, 378, 456):
        assert np.array_equal(o.viterbi_decode(g["vit%d_in" % n], n // 2), g["vit%d_out" % n]), n
        assert np.array_equal(o.encode(g["enc%d_in" % n]), g["enc%d_out" % n]), n
    for i in range(16):
        assert o.parity(fecbind.XCCH_POLY, 40, g["par_bits"][i, :184]) == int(g["par_xcch"][i])
        assert o.syndrome(fecbind.XCCH_POLY, 40, g["par_bits"][i]) == int(g["syn_xcch"][i])
        assert o.parity(fecbind.RACH_POLY, 6, g["par_bits"][i, :8]) == int(g["par_rach"][i])


def test_golden_xcch(o, golden):
    g = golden("fec.npz")
    for i in range(len(g["xcch_ok"])):
        r = o.xcch_decode(g["xcch_soft"][i])
        assert r["ok"] == bool(g["xcch_ok"][i]) and r["syndrome"] == int(g["xcch_syndrome"][i]), i
        assert np.array_equal(r["u"], g["xcch_u"][i]) and np.array_equal(r["d"], g["xcch_dout"][i]), i
        if r["ok"]:
            assert np.array_equal(r["d"], g["xcch_d"][i])          # the frame that was sent
    assert 30 < g["xcch_ok"].sum() < 60                            # both outcomes are exercised


def test_golden_rach(o, golden):
    g = golden("fec.npz")
    for i in range(len(g["rach_e"])):
        r = o.rach_decode(g["rach_e"][i])
        assert r["tail_ok"] == bool(g["rach_tail_ok"][i]) and r["bsic"] == int(g["rach_bsic_out"][i]), i
        assert r["ra"] == int(g["rach_ra_out"][i]) and np.array_equal(r["u"], g["rach_u"][i]), i


def test_golden_tch(o, golden):
    g = golden("fec.npz")
    for i in range(len(g["tch_good"])):
        r = o.tch_decode(g["tch_soft"][i])
        assert r["good"] == bool(g["tch_good"][i]), i
        assert np.array_equal(r["u"], g["tch_u"][i]) and np.array_equal(r["d"], g["tch_dout"][i]), i
    clean = np.flatnonzero(g["tch_good"] & (np.arange(len(g["tch_good"])) % 8 <= 1))
    assert len(clean) >= 12 and np.array_equal(g["tch_dout"][clean], g["tch_d"][clean])
    assert 0.3 < g["tch_good"].mean() < 0.9


def tch_bursts(rng, c_blocks):
    """[nblk,456] c[] per block -> [4*(nblk+1),148] bursts through the diagonal interleaver (GSM 05.03 3.1.3):
    block m sits in the even e-bits of bursts 4m..4m+3 and the odd e-bits of bursts 4m+4..4m+7."""
    nblk = c_blocks.shape[0]
    s = rng.random((4 * (nblk + 1), 148)).astype(np.float32)
    k = np.arange(456)
    j = 2 * ((49 * k) % 57) + ((k % 8) // 4)
    pos = np.where(j < 57, 3 + j, 88 + (j - 57))
    for m in range(nblk):
        s[4 * m + (k % 8), pos] = c_blocks[m]
    return s


def test_tch_batch_layout(o, golden):
    g = golden("fec.npz")
    rng = np.random.default_rng(8)
    b = tch_bursts(rng, g["tch_soft"])
    b[:, 60] = (np.arange(len(b)) % 3 == 0) * 0.9                  # Hl stealing flags
    r = o.tch_decode_batch(b, wire=False, nthreads=2)
    assert np.array_equal(r["good"].astype(bool), g["tch_good"])
    assert np.array_equal(np.unpackbits(r["tch"], axis=1)[:, :260], g["tch_dout"])
    assert np.array_equal(r["stolen"], ((4 * np.arange(len(r["stolen"])) + 7) % 3 == 0).astype(np.uint8))
    # the FACCH leg is the XCCH decode of the same c[]
    for m in (0, 5, 17):
        i4 = np.zeros((4, 114), np.float32)
        k = np.arange(456)
        i4[k % 4, 2 * ((49 * k) % 57) + ((k % 8) // 4)] = g["tch_soft"][m]
        x = o.xcch_decode(i4)
        assert bool(r["facch_ok"][m]) == x["ok"] and np.array_equal(np.unpackbits(r["facch"][m]), x["d"])


def test_batch_layout_and_wire(o, golden):
    """The batch forms read bursts as the transceiver delivers them (148 soft bits, e-bits at 3..59 and
    88..144, RACH payload at 49..84) and model the UDP hop's 8-bit quantisation."""
    g = golden("fec.npz")
    rng = np.random.default_rng(3)
    nb = len(g["xcch_ok"])
    soft = rng.random((4 * nb, 148)).astype(np.float32)
    x = g["xcch_soft"].reshape(4 * nb, 114)
    soft[:, 3:60] = x[:, :57]; soft[:, 88:145] = x[:, 57:]
    frames, ok = o.xcch_decode_batch(soft, wire=False, nthreads=2)
    assert np.array_equal(ok.astype(bool), g["xcch_ok"])
    good = np.flatnonzero(ok)
    assert np.array_equal(np.unpackbits(frames[good], axis=1), g["xcch_dout"][good])
    fw, okw = o.xcch_decode_batch(soft, wire=True, nthreads=2)
    for i in range(0, nb, 5):
        r = o.xcch_decode(o.wire(g["xcch_soft"][i]))
        assert bool(okw[i]) == r["ok"]
        assert np.array_equal(np.unpackbits(fw[i]), r["d"])
    v = np.float32(0.7004)
    assert o.wire(v) == np.float32(round(float(v) * 255.0) / 256.0)
    rs = rng.random((len(g["rach_e"]), 148)).astype(np.float32)
    rs[:, 49:85] = g["rach_e"]
    out = o.rach_decode_batch(rs, wire=False, nthreads=2)
    assert np.array_equal(out[:, 0].astype(bool), g["rach_tail_ok"])
    assert np.array_equal(out[:, 1], g["rach_bsic_out"]) and np.array_equal(out[:, 2], g["rach_ra_out"])


@pytest.mark.skipif(not reffec.available(), reason="oracle/_ref/libref_fec.so not built (reference absent)")
def test_random_vs_reference(o):
    r = reffec.RefFec()
    rng = np.random.default_rng(77)
    for it in range(250):
        d = rng.integers(0, 2, 184).astype(np.uint8)
        i4 = r.xcch_encode(d).astype(np.float32)
        s = np.clip(i4 * 0.8 + 0.1 + rng.normal(0, rng.choice([0.0, 0.1, 0.25, 0.4, 0.7]), i4.shape), 0, 1).astype(np.float32)
        if it % 3 == 0:
            s = o.wire(s)
        a, b = r.xcch_decode(s), o.xcch_decode(s)
        assert a["ok"] == b["ok"] and a["syndrome"] == b["syndrome"]
        assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["d"], b["d"])
        e = rng.random(36).astype(np.float32)
        a, b = r.rach_decode(e), o.rach_decode(e)
        assert (a["tail_ok"], a["bsic"], a["ra"]) == (b["tail_ok"], b["bsic"], b["ra"]) and np.array_equal(a["u"], b["u"])
        d = rng.integers(0, 2, 260).astype(np.uint8)
        s = np.clip(r.tch_encode(d) * 0.8 + 0.1 + rng.normal(0, rng.choice([0.0, 0.1, 0.3, 0.6]), 456), 0, 1).astype(np.float32)
        a, b = r.tch_decode(s), o.tch_decode(s)
        assert a["good"] == b["good"] and np.array_equal(a["u"], b["u"]) and np.array_equal(a["d"], b["d"])
        n = int(rng.integers(1, 230)) * 2
        sv = rng.random(n).astype(np.float32)
        if it % 4 == 0:
            sv[rng.integers(0, n, n // 3)] = 0.5                   # exact ties in the metrics
        assert np.array_equal(r.soft_decode(sv, n // 2), o.viterbi_decode(sv, n // 2))
        bits = rng.integers(0, 2, n // 2).astype(np.uint8)
        assert np.array_equal(r.encode(bits), o.encode(bits))
        assert r.parity(fecbind.XCCH_POLY, 40, 224, bits) == o.parity(fecbind.XCCH_POLY, 40, bits)
        assert r.syndrome(fecbind.RACH_POLY, 6, 8, bits) == o.syndrome(fecbind.RACH_POLY, 6, bits)
        assert np.array_equal(r.lsb8msb(bits), o.lsb8msb(bits))


def test_xcch_encode_golden(o, golden):
    """fo_xcch_encode against the e-bits the reference's encoder steps produced (golden xcch_d -> xcch_hard)."""
    g = golden("fec.npz")
    tsc = np.array([int(c) for c in "01001110101100000100111010"], np.uint8)        # TSC 5
    for i in range(0, len(g["xcch_d"]), 3):
        frame = np.packbits(g["xcch_d"][i])
        b = o.xcch_encode(frame, tsc)
        e = np.concatenate([b[:, 3:60], b[:, 88:145]], axis=1)
        assert np.array_equal(e, g["xcch_hard"][i]), i
        assert not b[:, :3].any() and not b[:, 145:].any() and b[:, 60].all() and b[:, 87].all()
        assert np.array_equal(b[:, 61:87], np.tile(tsc, (4, 1)))
