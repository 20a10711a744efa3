"""GPU parity for the TX / front-end rows: modulateBurst, polyphaseResampleVector, int16 <-> float.
Golden vectors from the real reference + random cases against the CPU oracle.  Value-exact."""
import numpy as np
import pytest

import _pkg
import oraclebind
from util import assert_veq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available()
    return _pkg.load()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = {s: pkg.TrxSig(s, 0) for s in (1, 2, 4)}
    for v in c.values():
        v.use_torch_stream()
    return c


@pytest.mark.parametrize("sps", [1, 4])
def test_golden_modulate(ctx, golden, sps):
    g = golden("modulate.npz")
    p = "sps%d_" % sps
    out, off, length = ctx[sps].modulate_host(g[p + "bits"], g[p + "guard"])
    assert_veq(off, g[p + "off"]); assert_veq(length, g[p + "len"])
    assert_veq(out, g[p + "x"], "modulateBurst")


@pytest.mark.parametrize("sps", [1, 2, 4])
def test_random_modulate_with_gain(ctx, sps):
    import torch
    rng = np.random.default_rng(31 + sps)
    B = 300
    o = oraclebind.Oracle(sps)
    bits = rng.integers(0, 256, (B, 148)).astype(np.uint8)      # only bit 0 counts (BitVector.cpp:54-63)
    guard = (8 + (np.arange(B) % 4 == 0)).astype(np.int32)
    gain = (10.0 ** (-rng.integers(0, 30, B) / 10.0)).astype(np.float32)
    length = (sps * (148 + guard)).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.int32)
    d_out = torch.zeros(int(length.sum()), 2, device="cuda")
    ctx[sps].modulate(torch.from_numpy(bits).cuda(), torch.from_numpy(guard).cuda(), d_out,
                      torch.from_numpy(off).cuda(), gain=torch.from_numpy(gain).cuda())
    torch.cuda.synchronize()
    out = d_out.cpu().numpy().view(np.complex64).ravel()
    for b in range(B):
        ref = o.scale_vector(o.modulate((bits[b] & 1).astype(np.int8), int(guard[b])), complex(gain[b], 0))
        assert_veq(out[off[b]:off[b] + length[b]], ref, "burst %d" % b)


def test_golden_resample(ctx, golden):
    import torch
    g = golden("resample.npz")
    t = ctx[4]
    cases = [("rx4_x", 260, 96, "lpf651_gain260", "rx4_y651"), ("rx4_x", 260, 96, "lpf961_gain260", "rx4_y961"),
             ("rx1_x", 65, 96, "lpf651_gain65", "rx1_y651"), ("tx4_x", 96, 260, "lpf651_gain96", "tx4_y")]
    for xin, P, Q, lpf, yout in cases:
        x = g[xin]; n = len(x)
        m = t.resample_out_len(n, P, Q)
        assert m == len(g[yout])
        S = 3                                                   # three identical streams, strided
        d_x = torch.from_numpy(np.tile(x.view(np.float32), S)).cuda()
        d_l = torch.from_numpy(g[lpf]).cuda()
        d_y = torch.zeros(S, m + 5, 2, device="cuda")
        t.resample(d_x, n, n, S, P, Q, d_l, d_y, m + 5)
        torch.cuda.synchronize()
        y = d_y.cpu().numpy().view(np.complex64).reshape(S, m + 5)
        for s in range(S):
            assert_veq(y[s, :m], g[yout], "%s stream %d" % (yout, s))
            assert not y[s, m:].any()


def test_random_resample_vs_oracle(ctx):
    import torch
    rng = np.random.default_rng(5)
    o = oraclebind.Oracle(4)
    for (n, P, Q, L) in [(1056, 260, 96, 961), (3000, 96, 260, 651), (17, 260, 96, 961), (100, 3, 2, 101), (64, 1, 1, 21)]:
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        lpf = rng.standard_normal(L).astype(np.float32)
        ref = o.polyphase_resample(x, P, Q, lpf)
        d_y = torch.zeros(len(ref), 2, device="cuda")
        ctx[4].resample(torch.from_numpy(x.view(np.float32)).cuda(), n, n, 1, P, Q, torch.from_numpy(lpf).cuda(),
                        d_y, len(ref))
        torch.cuda.synchronize()
        assert_veq(d_y.cpu().numpy().view(np.complex64).ravel(), ref, "resample %d %d/%d" % (n, P, Q))


def test_int16_conversions(ctx):
    import torch
    rng = np.random.default_rng(6)
    iq = rng.integers(-32768, 32768, 2 * 5000).astype(np.int16)
    d_iq = torch.from_numpy(iq).cuda()
    d_x = torch.zeros(5000, 2, device="cuda")
    ctx[4].unpack_int16(d_iq, 5000, d_x, swap_iq=True)          # non-SWLOOPBACK: I/Q flipped (radioInterface.cpp:101-112)
    torch.cuda.synchronize()
    x = d_x.cpu().numpy()
    assert_veq(x[:, 0], iq[1::2].astype(np.float32)); assert_veq(x[:, 1], iq[0::2].astype(np.float32))
    ctx[4].unpack_int16(d_iq, 5000, d_x, swap_iq=False)
    torch.cuda.synchronize()
    x = d_x.cpu().numpy()
    assert_veq(x[:, 0], iq[0::2].astype(np.float32))
    f = (rng.uniform(-30000, 30000, 2 * 4096)).astype(np.float32)
    d_o = torch.zeros(2 * 4096, dtype=torch.int16, device="cuda")
    ctx[4].pack_int16(torch.from_numpy(f).cuda(), 4096, d_o)     # (short) cast: truncation toward zero
    torch.cuda.synchronize()
    assert_veq(d_o.cpu().numpy(), np.trunc(f).astype(np.int16))
