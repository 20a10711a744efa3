"""CPU checks of the host-orchestration model (oracle/transceiver_model.py) that need no GPU: GSM::Time
arithmetic across the hyperframe wrap, the expectedCorrType schedule table, and the datagram formats of
TRXManager/README.TRXManager."""
import numpy as np

import oraclebind
import transceiver_model as tm


def test_time_arithmetic_wraps_like_gsm_time():
    H = tm.HYPERFRAME
    assert tm.fn_delta(5, H - 5) == 10 and tm.fn_delta(H - 5, 5) == -10
    assert tm.time_less((H - 1, 7), (0, 0)) and not tm.time_less((0, 0), (H - 1, 7))
    assert tm.time_less((10, 3), (10, 4)) and not tm.time_less((10, 4), (10, 4))


def test_time_arithmetic_matches_the_compiled_reference(golden):
    """The model's GSM::Time ordering and frame differences against vectors captured from the reference's own GSMCommon
    (tests/golden/gsm_time.npz, oracle/gen_golden.py:gen_gsm_time): this much of the host orchestration is pinned."""
    g = golden("gsm_time.npz")
    for a, b, less, greater, equal, minus in zip(g["a"], g["b"], g["less"], g["greater"], g["equal"], g["minus"]):
        a, b = (int(a[0]), int(a[1])), (int(b[0]), int(b[1]))
        assert tm.time_less(a, b) == bool(less), (a, b)
        # operator> is NOT the mirror of operator< exactly half a hyperframe apart (FNDelta is -H/2 both ways there): a > b
        # is "frames apart > 0", as the reference writes it (GSMCommon.h:431-435)
        assert ((a[1] > b[1]) if a[0] == b[0] else tm.fn_delta(a[0], b[0]) > 0) == bool(greater), (a, b)
        assert (a == b) == bool(equal)
        assert tm.fn_delta(a[0], b[0]) == int(minus), (a, b)
    assert int(g["less"].sum()) > 500 and int(g["greater"].sum()) > 500 and int(g["equal"].sum()) > 5


def test_schedule_table_and_datagrams():
    m = tm.TransceiverModel(oraclebind.Oracle(1))
    for ts, code in enumerate([tm.I, tm.II, tm.IV, tm.V, tm.VII, tm.LOOPBACK, tm.NONE, tm.VI]):
        assert m.control("CMD SETSLOT %d %d" % (ts, code)) == "RSP SETSLOT 0 %d %d" % (ts, code)
    assert m.filler_modulus == [26, 26, 51, 51, 102, 26, 26, 51]
    got = {ts: [m.expected_corr_type(ts, fn) for fn in range(102)] for ts in range(8)}
    assert set(got[0]) == {tm.TSC} and set(got[6]) == {tm.OFF}
    assert got[1][:4] == [tm.TSC, tm.IDLE, tm.TSC, tm.IDLE]
    assert [fn for fn in range(51) if got[2][fn] == tm.RACH] == [f for f in range(51) if f % 10 < 2]
    assert [fn for fn in range(51) if got[3][fn] == tm.RACH] == [4, 5] + list(range(14, 37)) + [45, 46]
    assert [fn for fn in range(51) if got[4][fn] == tm.IDLE] == [12, 13, 14]
    assert [fn for fn in range(51) if got[5][fn] == tm.IDLE] == [48, 49, 50]
    soft = np.linspace(0, 1, 148).astype(np.float32)
    d = m.encode_rx_datagram(5, 0x01020304, 37, -300, soft)
    assert len(d) == 158 and d[0] == 5 and d[1:5] == bytes([1, 2, 3, 4]) and d[5] == 37
    assert int.from_bytes(d[6:8], "big", signed=True) == -300
    assert d[8] == 0 and d[8 + 147] == 255 and d[156:] == b"\x00\x00"
    tx = bytes([3]) + (123456).to_bytes(4, "big") + bytes([0xF6]) + bytes([1, 0] * 74)
    assert m.decode_tx_datagram(tx)[:3] == (3, 123456, -10)


def test_tx_datagram_with_out_of_range_frame_number_is_rejected():
    """A frame number >= gHyperframe (e.g. bytes FF FF FF FF, which an int would read as negative) must not reach the
    filler table: the C-ABI codec calls it badly formatted.  No GPU needed: the codec is a pure host function."""
    import ctypes as C
    import _pkg
    L = _pkg.load().lib()
    L.trxsig_trx_decode_tx_datagram.argtypes = [C.c_char_p, C.c_int] + [C.POINTER(C.c_int)] * 3 + [C.c_char_p]
    tn, fn, rssi = C.c_int(), C.c_int(), C.c_int()
    bits = C.create_string_buffer(148)
    ok = bytes([3]) + (tm.HYPERFRAME - 1).to_bytes(4, "big") + bytes([0xF6]) + bytes([1, 0] * 74)
    assert L.trxsig_trx_decode_tx_datagram(ok, len(ok), C.byref(tn), C.byref(fn), C.byref(rssi), bits) == 0
    assert (tn.value, fn.value, rssi.value) == (3, tm.HYPERFRAME - 1, -10)
    for bad_fn in (b"\xff\xff\xff\xff", b"\x80\x00\x00\x00", tm.HYPERFRAME.to_bytes(4, "big")):
        bad = bytes([3]) + bad_fn + bytes([0xF6]) + bytes([1, 0] * 74)
        assert L.trxsig_trx_decode_tx_datagram(bad, len(bad), C.byref(tn), C.byref(fn), C.byref(rssi), bits) != 0
