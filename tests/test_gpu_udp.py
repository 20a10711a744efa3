"""The UDP contract end to end (TRXManager/README.TRXManager): the `transceiver` process on the Transceiver GROUP
(openbts-ttsou_amd/trxsig_transceiver_udp, software-loopback radio) against scripted peers playing the GSM core --
control commands and responses, clock indications, transmit datagrams in, receive datagrams out whose soft bits
are the transmitted bits; one ARFCN on the equalising leg, and eight ARFCNs on TransceiverManager's port plan
(TRXManager/TRXManager.cpp:44-54) with a malformed datagram (the core is reminded of the clock), a stale burst (it
lands in the filler table and is sent when its frame comes round again) and a stalled frame (the radio under-runs, the
transmit latency grows by a frame, Transceiver.cpp:697-703)."""
import os
import socket
import subprocess
import time

import numpy as np
import pytest

import _pkg
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_udp_loopback_session():
    import torch
    assert torch.cuda.is_available()
    exe = os.path.join(ROOT, "openbts-ttsou_amd", "trxsig_transceiver_udp")
    assert os.path.exists(exe), "run make -C openbts-ttsou_amd/csrc"
    B = 25700 + (os.getpid() % 500) * 4
    socks = {}
    for name, p in (("clock", B + 100), ("ctl", B + 101), ("data", B + 102)):
        s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        s.bind(("127.0.0.1", p)); s.settimeout(20.0)
        socks[name] = s
    proc = subprocess.Popen([exe, "--port", str(B), "--sps", "1", "--frames", "700", "--slot-us", "300", "--tsc-leg", "equalize"],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        msg, _ = socks["clock"].recvfrom(100)
        assert msg.startswith(b"IND CLOCK ") and msg.endswith(b"\x00")

        def cmd(c):
            socks["ctl"].sendto(c.encode() + b"\x00", ("127.0.0.1", B + 1))
            r, _ = socks["ctl"].recvfrom(100)
            assert r.endswith(b"\x00")
            return r[:-1].decode()

        assert cmd("CMD POWERON") == "RSP POWERON 1"                  # not tuned yet
        assert cmd("CMD RXTUNE 890000") == "RSP RXTUNE 0 890000"
        assert cmd("CMD TXTUNE 935000") == "RSP TXTUNE 0 935000"
        assert cmd("CMD SETTSC 6") == "RSP SETTSC 0 6"      # (TSC 5: the reference's 16-bit channel estimate leaks a sidelobe into the DFE)
        assert cmd("CMD SETSLOT 1 1") == "RSP SETSLOT 0 1 1"          # TCH/F: a normal burst every frame
        assert cmd("CMD POWERON") == "RSP POWERON 0"
        assert cmd("CMD SETPOWER 0") == "RSP SETPOWER 0 0"
        # the latest clock indication tells the core where the transceiver is
        socks["clock"].settimeout(0.05)
        fn_now = None
        while True:
            try:
                m, _ = socks["clock"].recvfrom(100)
                fn_now = int(m[:-1].split()[2])
            except socket.timeout:
                break
        assert fn_now is not None
        rng = np.random.default_rng(5)
        sent = {}
        for k in range(40):
            fn = fn_now + 60 + 3 * k
            bits = synth.normal_bits(rng, 1, 6)[0]
            sent[fn] = bits
            socks["data"].sendto(bytes([1]) + fn.to_bytes(4, "big") + bytes([0]) + bits.tobytes(), ("127.0.0.1", B + 2))
        got = {}
        socks["data"].settimeout(15.0)
        t0 = time.time()
        while len(got) < len(sent) and time.time() - t0 < 25:
            try:
                d, _ = socks["data"].recvfrom(200)
            except socket.timeout:
                break
            assert len(d) == 158
            tn, fn = d[0], int.from_bytes(d[1:5], "big")
            if tn == 1 and fn in sent:
                got[fn] = d
        assert len(got) >= 36, (len(got), len(sent))
        for fn, d in got.items():
            hard = (np.frombuffer(d[8:156], np.uint8) > 127).astype(np.uint8)
            assert np.array_equal(hard, sent[fn]), fn
            toa = int.from_bytes(d[6:8], "big", signed=True)
            rssi = d[5] - 256 if d[5] > 127 else d[5]
            assert abs(toa) <= 64 and -6 <= rssi <= 2                # loopback: on time, just above the 9450 reference level
    finally:
        try:
            out, err = proc.communicate(timeout=60)
        except subprocess.TimeoutExpired:
            proc.kill(); out, err = proc.communicate()
        for s in socks.values():
            s.close()
    assert proc.returncode == 0, (out, err)
    assert "rx bursts sent" in out



def test_udp_eight_arfcns():
    import re
    import torch
    assert torch.cuda.is_available()
    exe = os.path.join(ROOT, "openbts-ttsou_amd", "trxsig_transceiver_udp")
    N = 8
    B = 27000 + (os.getpid() % 200) * 40
    clock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM); clock.bind(("127.0.0.1", B + 100)); clock.settimeout(20.0)
    ctl, data = [], []
    for i in range(N):                                               # ARFCNManager's sockets (TRXManager.cpp:123-124)
        c = socket.socket(socket.AF_INET, socket.SOCK_DGRAM); c.bind(("127.0.0.1", B + 101 + 2 * i)); c.settimeout(20.0); ctl.append(c)
        d = socket.socket(socket.AF_INET, socket.SOCK_DGRAM); d.bind(("127.0.0.1", B + 102 + 2 * i)); d.settimeout(0.02); data.append(d)
    # the core reads its data sockets all the time (ARFCNManager's receive loop, TRXManager.cpp:205-234): a reader thread.  (It matters:
    # with TSC 5 the reference's analyzeTrafficBurst "detects" the looped-back DUMMY burst of every idle frame at a TOA of 15.7
    # symbols -- identically in the compiled reference -- so that ARFCN's socket sees two datagrams per frame from POWERON on.)
    import select
    import threading
    inbox, stop = [], threading.Event()

    def reader():
        while not stop.is_set():
            r, _, _ = select.select(data, [], [], 0.05)
            for sk in r:
                try:
                    inbox.append((data.index(sk), sk.recvfrom(200)[0]))
                except (socket.timeout, OSError):
                    pass
    rd = threading.Thread(target=reader, daemon=True); rd.start()
    frame_us = 2500
    proc = subprocess.Popen([exe, "--port", str(B), "--arfcns", str(N), "--sps", "1", "--tsc-leg", "demod", "--frame-us", str(frame_us),
                             "--frames", "1500", "--stall-frame", "500", "--stall-ms", "40"] + (["--debug-arfcn", os.environ["TRXSIG_UDP_DEBUG"]] if "TRXSIG_UDP_DEBUG" in os.environ else []),
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def clocks(wait):
        """every clock indication that arrives within `wait` seconds -> list of frame numbers"""
        out = []
        clock.settimeout(wait)
        while True:
            try:
                m, _ = clock.recvfrom(100)
            except socket.timeout:
                return out
            assert m.startswith(b"IND CLOCK ") and m.endswith(b"\x00")
            out.append(int(m[:-1].split()[2]))
            clock.settimeout(0.02)

    try:
        assert clocks(20.0)                                          # the process is up

        def cmd(i, c):
            ctl[i].sendto(c.encode() + b"\x00", ("127.0.0.1", B + 1 + 2 * i))
            r, _ = ctl[i].recvfrom(100)
            assert r.endswith(b"\x00")
            return r[:-1].decode()

        slots = {}
        for i in range(N):
            assert cmd(i, "CMD RXTUNE %d" % (890000 + 200 * i)) == "RSP RXTUNE 0 %d" % (890000 + 200 * i)
            assert cmd(i, "CMD TXTUNE %d" % (935000 + 200 * i)) == "RSP TXTUNE 0 %d" % (935000 + 200 * i)
            assert cmd(i, "CMD SETTSC %d" % i) == "RSP SETTSC 0 %d" % i
            slots[i] = (i % 8, (i + 3) % 8)
            for tn in slots[i]:
                assert cmd(i, "CMD SETSLOT %d 1" % tn) == "RSP SETSLOT 0 %d 1" % tn
            assert cmd(i, "CMD POWERON") == "RSP POWERON 0"
        fn_now = clocks(0.1)[-1]                                     # (every command is answered with a clock indication too, :463)
        rng = np.random.default_rng(8)
        sent = {}
        for k in range(30):
            for i in range(N):
                tn = slots[i][k % 2]
                fn = fn_now + 80 + 2 * k
                bits = synth.normal_bits(rng, 1, i)[0]
                sent[(i, tn, fn)] = bits
                data[i].sendto(bytes([tn]) + fn.to_bytes(4, "big") + bytes([0]) + bits.tobytes(), ("127.0.0.1", B + 2 + 2 * i))
        # a stale burst: its frame has gone by -- it must land in the filler table [FN % 26][TN] and go out when that entry comes round
        stale_bits = synth.normal_bits(rng, 1, 5)[0]
        stale_fn = fn_now - 40
        data[5].sendto(bytes([slots[5][0]]) + stale_fn.to_bytes(4, "big") + bytes([0]) + stale_bits.tobytes(), ("127.0.0.1", B + 2 + 2 * 5))
        # a malformed datagram, then a good one: the core is reminded of the clock (:778-793)
        clocks(0.05)
        data[3].sendto(b"\x01" * 100, ("127.0.0.1", B + 2 + 2 * 3))
        junk_fn = fn_now + 200
        jb = synth.normal_bits(rng, 1, 3)[0]
        sent[(3, slots[3][0], junk_fn)] = jb
        data[3].sendto(bytes([slots[3][0]]) + junk_fn.to_bytes(4, "big") + bytes([0]) + jb.tobytes(), ("127.0.0.1", B + 2 + 2 * 3))
        assert clocks(0.3), "no clock indication after a malformed datagram"
        got, stale_seen, seen = {}, 0, 0
        t0 = time.time()
        while (len(got) < len(sent) or not stale_seen) and time.time() - t0 < 20:
            time.sleep(0.05)
            while seen < len(inbox):
                i, d = inbox[seen]; seen += 1
                assert len(d) == 158
                tn, fn = d[0], int.from_bytes(d[1:5], "big")
                hard = (np.frombuffer(d[8:156], np.uint8) > 127).astype(np.uint8)
                if (i, tn, fn) in sent:
                    got[(i, tn, fn)] = hard
                elif i == 5 and tn == slots[5][0] and (fn - stale_fn) % 26 == 0 and np.array_equal(hard, stale_bits):
                    stale_seen += 1
        missing = sorted((k[0], k[1], k[2] - fn_now) for k in sent if k not in got)
        print("fn_now", fn_now, "missing", missing, "stale_seen", stale_seen)
        assert len(got) >= len(sent) - 8, (len(got), len(sent), missing)
        for key, hard in got.items():
            assert np.array_equal(hard, sent[key]), key
        assert {k[0] for k in got} == set(range(N))                  # every ARFCN's bits came back
        assert stale_seen >= 1
    finally:
        stop.set(); rd.join(timeout=2)
        try:
            out, err = proc.communicate(timeout=60)
        except subprocess.TimeoutExpired:
            proc.kill(); out, err = proc.communicate()
        for s_ in [clock] + ctl + data:
            s_.close()
        if "TRXSIG_UDP_DEBUG" in os.environ:
            print(err)
    assert proc.returncode == 0, (out, err)
    m = re.search(r"malformed (\d+) .*under-runs (\d+)  transmit latency (\d+):(\d+)  service time per frame avg ([0-9.]+) us", out)
    assert m, out
    assert int(m.group(1)) == 1 and int(m.group(2)) >= 1             # the stalled frame under-ran the radio ...
    assert (int(m.group(3)), int(m.group(4))) != (2, 0)              # ... and the latency controller answered (:697-714)
    assert float(m.group(5)) < frame_us, out                         # eight ARFCNs' frame is served well inside the frame time
    print(out)
