"""driveTransmitFIFO's deadline clock / latency controller and writeClockInterface (Transceiver/Transceiver.cpp:679-746) as the
library exposes them (trxsig_txclock_*, include/trxsig_transceiver.h): a pure state machine, so this runs without a GPU.
  * its GSM::Time arithmetic (operator+(Time), incTN, decTN, operator>) is held against vectors captured from the reference's own
    class (tests/golden/gsm_time.npz, compiled from GSM/GSMCommon.h by oracle/ref_driver.cpp);
  * the controller against a line-by-line Python restatement of :679-729 on long random radio-clock walks with under-runs
    (restatement: unpinned, as all of Transceiver.cpp -- it cannot be compiled here)."""
import ctypes as C

import numpy as np
import pytest

import _pkg
import transceiver_model as tm

H = tm.HYPERFRAME


class TxClock(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("deadline_fn", "deadline_tn", "latency_fn", "latency_tn", "latency_update_fn",
                                       "latency_update_tn", "last_clock_fn", "last_clock_tn")]


@pytest.fixture(scope="module")
def L():
    lib = _pkg.load().lib()
    lib.trxsig_txclock_init.argtypes = [C.POINTER(TxClock)] + [C.c_int] * 4
    lib.trxsig_txclock_init.restype = None
    lib.trxsig_txclock_advance.argtypes = [C.POINTER(TxClock), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.trxsig_txclock_indication_due.argtypes = [C.POINTER(TxClock)]
    lib.trxsig_txclock_indication.argtypes = [C.POINTER(TxClock), C.c_char_p, C.c_int]
    return lib


def plus(a, b):                                              # Time::operator+(const Time&) (GSMCommon.h:405-410)
    return ((a[0] + b[0] + (a[1] + b[1]) // 8) % H, (a[1] + b[1]) % 8)


def inc_tn(t):
    return (t[0], t[1] + 1) if t[1] < 7 else ((t[0] + 1) % H, 0)


def dec_tn(t):
    return (t[0], t[1] - 1) if t[1] > 0 else ((t[0] - 1) % H, 7)


def test_time_arithmetic_of_the_controller_is_the_reference_class(L, golden):
    """One loop iteration with latency (0, 0)... cannot isolate operator+ from outside, so: deadline = a, radio = b, latency chosen
    so that exactly the golden sum decides -- the controller pushes iff (b + latency) > a, with b + latency from the golden file."""
    g = golden("gsm_time.npz")
    n_push = 0
    for a, b, p in zip(g["a"][:600], g["b"][:600], g["plus"][:600]):
        a, b, p = tuple(int(v) for v in a), tuple(int(v) for v in b), tuple(int(v) for v in p)
        assert plus(a, b) == p                               # the restatement used below is the reference's operator+
        c = TxClock()
        L.trxsig_txclock_init(C.byref(c), 5, 3, b[0], b[1])  # deadline (5, 3), latency = b
        fn, tn, ur = C.c_int(), C.c_int(), C.c_int(0)
        n = L.trxsig_txclock_advance(C.byref(c), a[0], a[1], C.byref(ur), 1, C.byref(fn), C.byref(tn))
        want = tm.time_greater(p, (5, 3))                    # radio a + latency b > deadline ?
        assert n == int(want) and (fn.value, tn.value) == (5, 3)
        assert (c.deadline_fn, c.deadline_tn) == (inc_tn((5, 3)) if want else (5, 3))
        n_push += n
    assert 100 < n_push < 600
    sel = np.flatnonzero(g["tn_step"] == 1)                  # incTN() / decTN() as the restatement below uses them
    assert len(sel) > 50
    for i in sel:
        t = tuple(int(v) for v in g["a"][i])
        assert inc_tn(t) == tuple(int(v) for v in g["inc_tn"][i]) and dec_tn(t) == tuple(int(v) for v in g["dec_tn"][i])


def model_advance(s, radio, underrun):
    """Transceiver.cpp:689-718 on s = dict(deadline, latency, upd); returns (slots pushed, underrun flag left)."""
    n = 0
    while tm.time_greater(plus(radio, s["latency"]), s["deadline"]):
        if underrun:                                         # isUnderrun() (read and clear)
            underrun = False
            if tm.time_greater(radio, plus(s["upd"], (10, 0))):
                s["latency"] = plus(s["latency"], (1, 0)); s["upd"] = radio
        elif tm.time_greater(s["latency"], (1, 1)):
            if tm.time_greater(radio, plus(s["upd"], (216, 0))):
                s["latency"] = dec_tn(s["latency"]); s["upd"] = radio
        s["deadline"] = inc_tn(s["deadline"])
        n += 1
    return n, underrun


@pytest.mark.parametrize("seed,start", [(1, (2, 0)), (2, (H - 300, 5)), (3, (777777, 3))])
def test_latency_controller_follows_the_restatement(L, seed, start):
    rng = np.random.default_rng(seed)
    c = TxClock()
    L.trxsig_txclock_init(C.byref(c), start[0], start[1], 2, 0)          # runTransceiver.cpp:53: GSM::Time(2,0)
    s = {"deadline": start, "latency": (2, 0), "upd": start}
    last_clock = start
    radio = start
    grew = shrank = pushed = inds = 0
    for step in range(6000):
        for _ in range(int(rng.integers(1, 12))):                        # the radio clock moves on by some timeslots
            radio = inc_tn(radio)
        ur = bool(rng.random() < (0.02 if step < 3000 else 0.0))         # under-runs early on, then a long quiet stretch
        before = s["latency"]
        want_n, _ = model_advance(s, radio, ur)
        got_n = 0
        first = None
        flag = C.c_int(int(ur))
        while True:                                                      # in pieces of at most 5 timeslots
            fn, tn = C.c_int(), C.c_int()
            k = L.trxsig_txclock_advance(C.byref(c), radio[0], radio[1], C.byref(flag), 5, C.byref(fn), C.byref(tn))
            assert k >= 0
            if first is None:
                first = (fn.value, tn.value)
            got_n += k
            if k < 5:
                break
        assert got_n == want_n and (c.deadline_fn, c.deadline_tn) == s["deadline"]
        assert (c.latency_fn, c.latency_tn) == s["latency"] and (c.latency_update_fn, c.latency_update_tn) == s["upd"]
        grew += s["latency"][0] > before[0]; shrank += tm.time_greater(before, s["latency"]) and s["latency"][0] <= before[0]
        pushed += got_n
        due = tm.time_greater(s["deadline"], plus(last_clock, (216, 0)))
        assert L.trxsig_txclock_indication_due(C.byref(c)) == int(due)
        if due:
            buf = C.create_string_buffer(64)
            n = L.trxsig_txclock_indication(C.byref(c), buf, 64)
            assert buf.value.decode() == "IND CLOCK %d" % (s["deadline"][0] + 20) and n == len(buf.value)
            last_clock = s["deadline"]; inds += 1
    assert grew >= 3 and shrank >= 3 and pushed > 30000 and inds >= 10
