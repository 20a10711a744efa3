"""TX chain end to end on the GPU (pushRadioVector side): bursts -> modulateBurst ->
send buffer with history -> polyphaseResampleVector(96 : 65*sps, sendLPF) -> scaleVector(13500) ->
USRPifyVector int16, against the same chain built from the CPU oracle.  Value-exact, in both forms of the back end: fused
(one kernel per pop that modulates from the queued bits; the complex float32 send buffer never exists) and unfused."""
import numpy as np
import pytest

import _pkg
import oraclebind

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("sps", [1, 4])
def test_tx_chain(golden, sps, fused):
    import torch
    assert torch.cuda.is_available()
    pkg = _pkg.load()
    from openbts_ttsou_amd.frontend import TxBackEnd, OUTHISTORY
    from openbts_ttsou_amd import synth
    S = 3
    g = golden("resample.npz")
    lpf = g["lpf651_gain96"]                                    # createLPF(1/260, 651, P=96) as pushBuffer builds it
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    be = TxBackEnd(ctx, S, lpf, fused=fused)      # fused: modulate -> resample -> int16 in one kernel, no complex float32 send buffer
    o = oraclebind.Oracle(sps)
    rng = np.random.default_rng(31 + sps)
    hist = [np.zeros(2 * 65 * sps, np.complex64) for _ in range(S)]
    send = [np.zeros(0, np.complex64) for _ in range(S)]
    inchunk = 65 * 9 * sps
    tn = 0
    nout = 0
    for it in range(14):
        nb = int(rng.integers(1, 6)) if it != 9 else 40       # (one push of many bursts: several chunks in one pop)
        guard = np.array([8 + ((tn + k) % 4 == 0) for k in range(nb)], np.int32); tn = (tn + nb) % 8
        bits = np.stack([synth.normal_bits(rng, nb, int(rng.integers(0, 8))) for _ in range(S)])    # [S, nb, 148]
        gain = rng.uniform(0.1, 1.0, (S, nb)).astype(np.float32) if it % 2 else None
        be.push_bursts(bits, guard, gain)
        got = be.pop_samples()
        for s in range(S):
            for k in range(nb):
                x = o.modulate(bits[s, k].astype(np.int8), int(guard[k]))
                if gain is not None:
                    x = o.scale_vector(x, complex(gain[s, k], 0.0))
                send[s] = np.concatenate([send[s], x])
        nch = len(send[0]) // inchunk
        if nch == 0:
            assert got is None
            continue
        iq = got.cpu().numpy()
        for s in range(S):
            tr = send[s][:nch * inchunk]
            y = o.polyphase_resample(np.concatenate([hist[s], tr]), 96, 65 * sps, lpf)
            y = o.scale_vector(y, complex(13500.0, 0.0))
            want = np.stack([np.trunc(y.real), np.trunc(y.imag)], axis=1).astype(np.int16)[OUTHISTORY:]
            assert iq[s].shape == want.shape, (iq[s].shape, want.shape)
            assert np.array_equal(iq[s], want), (it, s)
            hist[s] = tr[-2 * 65 * sps:]
            send[s] = send[s][nch * inchunk:]
        nout += iq.shape[1]
    assert nout > 3000
