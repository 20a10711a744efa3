"""The host-side Transceiver orchestration (include/trxsig_transceiver.h) against
oracle/transceiver_model.py, the restatement of Transceiver/Transceiver.cpp on the CPU oracle: a scripted run of
several hundred bursts over mixed slot configurations -- normal bursts through the cached-DFE equaliser leg,
access bursts, noise (false detections raise the threshold), silence (the threshold decays) -- compared burst by
burst: what comes back, soft bits, RSSI, timing offset and the adaptive threshold (exact double equality);
then the transmit queue / filler table, the control commands and the UDP datagram codecs."""
import numpy as np
import pytest

import _pkg
import oraclebind
import synth
import transceiver_model as tm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _pkg.load()


def configure(x):
    r = [x.control("CMD RXTUNE 890000"), x.control("CMD TXTUNE 935000"), x.control("CMD SETTSC 5"),
         x.control("CMD SETSLOT 0 5"), x.control("CMD SETSLOT 1 1"), x.control("CMD SETSLOT 2 7"),
         x.control("CMD SETSLOT 3 2"), x.control("CMD SETSLOT 4 4"), x.control("CMD POWERON")]
    return r


def test_receive_state_machine(pkg):
    sps = 1
    o = oraclebind.Oracle(sps)
    h = pkg.TrxHost(sps, 0, start=(100, 0)); m = tm.TransceiverModel(o, start=(100, 0))
    assert configure(h) == configure(m)
    rng = np.random.default_rng(7)
    nb, nr = 400, 160
    xs, offs, lens, meta = synth.normal_batch(sps, nb, 5, seed=3, sigmas=(0.0, 0.05, 0.2, 0.6), max_delay=1.2)
    xr, offr, lenr, _ = synth.rach_batch(sps, nr, seed=4, sigmas=(0.0, 0.1, 0.3), max_delay_sym=20)
    # two-path channel on a third of the normal bursts (gives the DFE something to do)
    for i in range(0, nb, 3):
        s = xs[offs[i]:offs[i] + lens[i]]
        s[1:] = s[1:] + np.complex64(0.35 - 0.2j) * s[:-1].copy()
    ib = ir = 0
    outcomes = {"none": 0, "tsc": 0, "rach": 0}
    thr_path = []
    fn = 100
    for step in range(900):
        tn = step % 5
        if tn == 0:
            fn += 1
        if 300 <= step < 420:                                            # a long quiet spell
            fn += 3
        ct = m.expected_corr_type(tn, fn)
        assert h.expected_corr_type(tn, fn) == ct
        kind = rng.integers(0, 10)
        n = 156 + (tn % 4 == 0)
        if 300 <= step < 420 or kind == 0:
            x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * np.float32(0.5)       # silence
        elif kind <= 2:
            x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * rng.uniform(300, 2500)).astype(np.complex64)   # loud noise
        elif ct == tm.RACH and ir < nr:
            x = xr[offr[ir]:offr[ir] + lenr[ir]][:n].copy(); ir += 1
            if len(x) < n:
                x = np.concatenate([x, np.zeros(n - len(x), np.complex64)])
        else:
            x = xs[offs[ib]:offs[ib] + lens[ib]][:n].copy(); ib = (ib + 1) % nb
            if len(x) < n:
                x = np.concatenate([x, np.zeros(n - len(x), np.complex64)])
        a = h.pull_radio_vector(x, tn, fn)
        b = m.pull_radio_vector(x, tn, fn)
        assert h.energy_threshold == m.energy_threshold, (step, h.energy_threshold, m.energy_threshold)
        thr_path.append(m.energy_threshold)
        assert (a is None) == (b is None), (step, tn, fn, ct)
        if a is None:
            outcomes["none"] += 1
            continue
        outcomes["tsc" if ct == tm.TSC else "rach"] += 1
        assert a[1] == b[1] and a[2] == b[2], (step, a[1:], b[1:])
        assert np.array_equal(a[0][:148], b[0][:148]), (step, ct)
        assert h.encode_rx_datagram(tn, fn, a[1], a[2], a[0]) == m.encode_rx_datagram(tn, fn, b[1], b[2], b[0])
    assert outcomes["tsc"] > 150 and outcomes["rach"] > 20 and outcomes["none"] > 200, outcomes
    assert min(thr_path) < 200 and max(thr_path) > 255                  # the threshold really moved both ways
    h.close()


def test_transmit_queue_filler_and_codecs(pkg, golden):
    sps = 4
    o = oraclebind.Oracle(sps)
    h = pkg.TrxHost(sps, 0); m = tm.TransceiverModel(o)
    for x in (h, m):
        x.control("CMD SETSLOT 0 5"); x.control("CMD SETSLOT 1 1"); x.control("CMD SETSLOT 2 7")
    assert [h.filler_modulus(t) for t in range(4)] == [51, 26, 102, 26] == m.filler_modulus[:4]
    rng = np.random.default_rng(9)
    # the filler table starts as the modulated dummy burst
    a, fq = h.push_radio_vector(3, 7); b, fqm = m.push_radio_vector(3, 7)
    assert not fq and not fqm and np.array_equal(a, b)
    assert np.array_equal(np.array([int(c) for c in tm.DUMMY_BURST], np.int8), golden("tables.npz")["dummy_burst"].astype(np.int8))
    # bursts arrive out of order, some stale, some for the future; the datagram codec feeds addRadioVector
    sent = []
    for k in range(60):
        tn, fn, rssi = int(rng.integers(0, 3)), int(rng.integers(20, 60)), int(rng.integers(-5, 47))
        bits = rng.integers(0, 2, 148).astype(np.uint8)
        dg = bytes([tn]) + fn.to_bytes(4, "big") + bytes([rssi & 0xff]) + bits.tobytes()
        d1, d2 = h.decode_tx_datagram(dg), m.decode_tx_datagram(dg)
        assert d1[:3] == d2[:3] == (tn, fn, rssi) and np.array_equal(d1[3], d2[3])
        h.add_radio_vector(d1[3], d1[2], d1[0], d1[1]); m.add_radio_vector(d2[3], d2[2], d2[0], d2[1])
        sent.append((fn, tn))
    assert h.decode_tx_datagram(b"\x00" * 153) is None and m.decode_tx_datagram(b"\x00" * 153) is None
    assert h.queue_size() == len(m.queue) == 60
    nq = 0
    for fn in range(30, 70):
        for tn in range(3):
            a, fq = h.push_radio_vector(tn, fn); b, fqm = m.push_radio_vector(tn, fn)
            assert fq == fqm and np.array_equal(a, b), (fn, tn)
            nq += fq
    assert nq >= 20 and h.queue_size() == len(m.queue) == 0
    # a later frame with the same FN modulus replays what was stored in the filler table
    a, _ = h.push_radio_vector(1, 30 + 26 * 5); b, _ = m.push_radio_vector(1, 30 + 26 * 5)
    assert np.array_equal(a, b)
    # a frame number outside [0, gHyperframe) never reaches the queue or the filler table
    import pytest
    bits = np.zeros(148, np.uint8)
    for bad in (-1, -(2 ** 31), tm.HYPERFRAME):
        with pytest.raises(pkg.TrxSigError):
            h.add_radio_vector(bits, 0, 1, bad)
        with pytest.raises(pkg.TrxSigError):
            h.push_radio_vector(1, bad)
    assert h.queue_size() == 0
    assert h.decode_tx_datagram(bytes([1]) + b"\xff\xff\xff\xff" + bytes(149)) is None
    # createLPF normalisation (a21)
    g = golden("resample.npz")
    assert np.array_equal(h.create_lpf(g["rcvLPF_651_raw"], 96.0), g["lpf651_gain96"])
    assert np.array_equal(h.create_lpf(g["sendLPF_961_raw"], 260.0), g["lpf961_gain260"])
    h.close()


def test_control_commands(pkg):
    o = oraclebind.Oracle(1)
    h = pkg.TrxHost(1, 0); m = tm.TransceiverModel(o)
    script = ["CMD POWERON", "CMD SETPOWER 5", "CMD ADJPOWER 3", "CMD RXTUNE 890200", "CMD POWERON", "CMD TXTUNE 935200",
              "CMD SETTSC 3", "CMD SETTSC 9", "CMD SETTSC -1", "CMD SETMAXDELAY 3", "CMD SETSLOT 3 7", "CMD SETSLOT 9 1",
              "CMD POWERON", "CMD POWERON", "CMD SETPOWER 7", "CMD SETMAXDELAY 2",
              "CMD ADJPOWER -4", "CMD RXTUNE 1", "CMD TXTUNE 2", "CMD SETTSC 2", "CMD POWEROFF", "XYZ POWERON", "CMD NOSUCH 1",
              "CMD SETSLOT 0 4", "CMD SETPOWER", "CMD SETSLOT -1 1"]
    for c in script:
        assert h.control(c) == m.control(c), c
    assert h.control("CMD ADJPOWER 0") == "RSP ADJPOWER 0 0"      # ("CMD SETPOWER" with its integer missing reads as 0)
    assert h.control("CMD SETTSC 9") == "RSP SETTSC 1 9"          # out of range: refused by product and model alike
    h.close()
