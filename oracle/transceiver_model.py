"""Restatement of the reference's per-ARFCN host orchestration -- class Transceiver,
Transceiver/Transceiver.cpp -- on top of the CPU oracle's sigProcLib restatement.  TEST INFRASTRUCTURE ONLY:
the checker for include/trxsig_transceiver.h.

Parity status: UNPINNED for the orchestration itself.  Transceiver.cpp cannot be compiled here (it pulls in
radioInterface.h / the USRP driver headers, which this image lacks), and the reference ships no test or vector
for it, so this file is a line-by-line restatement from the source; every sigProcLib call inside it goes to
oracle/sigproc_oracle.c, which IS pinned.  Line numbers: Transceiver/Transceiver.cpp."""
import heapq
import math

import numpy as np

HYPERFRAME = 2048 * 26 * 51
OFF, TSC, RACH, IDLE = range(4)
NONE, I, II, III, IV, V, VI, VII, LOOPBACK = range(9)
DUMMY_BURST = ("0001111101101110110000010100100111000001001000100000001111100011100010111000101110001010111010010100"
               "011001100111001111010011111000100101111101010000")


def fn_delta(v1, v2):                       # GSM/GSMCommon.cpp:161-168
    half = HYPERFRAME // 2
    d = v1 - v2
    if d >= half:
        d -= HYPERFRAME
    elif d < -half:
        d += HYPERFRAME
    return d


def time_less(a, b):                        # GSM::Time::operator< (GSMCommon.h:425-429); a, b = (fn, tn)
    if a[0] == b[0]:
        return a[1] < b[1]
    return fn_delta(a[0], b[0]) < 0


def time_greater(a, b):                     # GSM::Time::operator> (GSMCommon.h:431-435)
    if a[0] == b[0]:
        return a[1] > b[1]
    return fn_delta(a[0], b[0]) > 0


class Queued:
    """One entry of mTransmitPriorityQueue.  The reference's queue is a std::priority_queue<radioVector*, std::vector<...>,
    PointerCompare> with the comparator *v1 > *v2 on the bursts' timestamps (CommonLibs/Interthread.h:432-463,
    Transceiver/radioInterface.h:58, 64-72); which of two bursts with EQUAL timestamps leaves first depends on the heap's
    shape.  Python's heapq moves elements exactly as libstdc++'s push_heap / pop_heap do (sift to a leaf taking the right child
    on a tie, then up while strictly smaller), so with __lt__(a, b) = comp(b, a) the model's queue has the same shape
    (tests/test_txqueue_order.py holds both against std::priority_queue itself)."""
    __slots__ = ("time", "payload")

    def __init__(self, time, payload):
        self.time, self.payload = time, payload

    def __lt__(self, other):
        return time_greater(other.time, self.time)


class TransceiverModel:
    def __init__(self, oracle, start=(0, 0), need_dfe=True):
        """need_dfe=False: the TSC leg as Transceiver52M/Transceiver.cpp runs it while mMaxExpectedDelay <= 1
        (:272 needDFE false -> :322 no channel estimate, :382 demodulateBurst instead of equalizeBurst); the sigProcLib
        calls stay Transceiver/sigProcLib.cpp's and so does the slot schedule."""
        self.need_dfe = need_dfe
        self.o = oracle
        self.sps = oracle.sps
        self.on = False; self.tx_freq = 0.0; self.rx_freq = 0.0; self.power = -10; self.tsc = 0
        self.chan_type = [NONE] * 8
        self.energy_threshold = 250.0                                        # :88 (double)
        self.prev_false = start
        self.chan = [None] * 8                                               # per slot: (w, b, chan_off)
        self.est_time = [start] * 8
        self.filler_modulus = [26] * 8
        dummy = np.array([int(c) for c in DUMMY_BURST], np.int8)
        self.filler = [[None] * 8 for _ in range(102)]
        for i in range(8):                                                   # :68-85
            m = self.o.modulate(dummy, 8 + (i % 4 == 0))
            for j in range(102):
                self.filler[j][i] = m
        self.queue = []                                                      # [(time, samples)], earliest first

    # ---- control (:439-580) ----
    def control(self, msg):
        parts = msg.split()
        if len(parts) < 2 or parts[0][:3] != "CMD":
            return ""
        cmd = parts[1]
        arg = lambda k: int(parts[2 + k]) if len(parts) > 2 + k else 0      # (a missing integer reads as 0 in the product)
        if cmd == "POWEROFF":
            return "RSP POWEROFF 0"
        if cmd == "POWERON":
            if not self.tx_freq or not self.rx_freq:
                return "RSP POWERON 1"
            if not self.on:
                self.power = -20; self.on = True
            return "RSP POWERON 0"
        if cmd == "SETPOWER":
            if not self.on:
                return "RSP SETPOWER 1 %d" % arg(0)
            self.power = arg(0)
            return "RSP SETPOWER 0 %d" % arg(0)
        if cmd == "ADJPOWER":
            if not self.on:
                return "RSP ADJPOWER 1 %d" % self.power
            self.power += arg(0)
            return "RSP ADJPOWER 0 %d" % self.power
        if cmd in ("RXTUNE", "TXTUNE"):
            if self.on:
                return "RSP %s 1 %d" % (cmd, arg(0))
            if cmd == "RXTUNE":
                self.rx_freq = arg(0) * 1.0e3
            else:
                self.tx_freq = arg(0) * 1.0e3
            return "RSP %s 0 %d" % (cmd, arg(0))
        if cmd == "SETTSC":
            # (a TSC outside 0..7: the reference stores it and then indexes gMidambles[] with it, sigProcLib.cpp:946 -- undefined
            #  behaviour; the product refuses it with status 1 and so does this model)
            if self.on or not 0 <= arg(0) <= 7:
                return "RSP SETTSC 1 %d" % arg(0)
            self.tsc = arg(0)
            return "RSP SETTSC 0 %d" % arg(0)
        if cmd == "SETMAXDELAY":                                             # Transceiver52M/Transceiver.cpp:476-486
            if not self.on:
                return "RSP SETMAXDELAY 1 %d" % arg(0)
            self.max_delay = arg(0)
            return "RSP SETMAXDELAY 0 %d" % arg(0)
        if cmd == "SETSLOT":
            ts, code = arg(0), arg(1)
            if ts < 0 or ts > 7:
                return ""
            self.chan_type[ts] = code
            if code in (NONE, I, II, III):                                   # setModulus (:183-204)
                self.filler_modulus[ts] = 26
            elif code in (IV, VI, V):
                self.filler_modulus[ts] = 51
            elif code == VII:
                self.filler_modulus[ts] = 102
            return "RSP SETSLOT 0 %d %d" % (ts, code)
        return ""

    def expected_corr_type(self, tn, fn):                                    # :207-269
        ct = self.chan_type[tn]
        if ct == NONE: return OFF
        if ct == I: return TSC
        if ct == II: return IDLE if fn % 2 == 1 else TSC
        if ct == III: return TSC
        if ct in (IV, VI): return RACH if (fn % 51) % 10 < 2 else OFF
        if ct == V:
            m = fn % 51
            if 14 <= m <= 36 or m in (4, 5, 45, 46): return RACH
            return TSC
        if ct == VII: return IDLE if fn % 51 in (12, 13, 14) else TSC
        if ct == LOOPBACK: return IDLE if 48 <= fn % 51 <= 50 else TSC
        return OFF

    # ---- pullRadioVector (:271-410) ----
    def pull_radio_vector(self, x, tn, fn):
        now = (fn, tn)
        ct = self.expected_corr_type(tn, fn)
        if ct in (OFF, IDLE):
            return None
        ok_e, _ = self.o.energy_detect(x, 20 * self.sps, np.float32(self.energy_threshold))
        if not ok_e:
            if float(fn_delta(fn, self.prev_false[0])) > 50:
                self.energy_threshold -= 10.0
                self.prev_false = now
            return None
        if ct == TSC:
            estimate = float(fn_delta(fn, self.est_time[tn][0])) > 50 or self.chan[tn] is None
            if estimate:
                self.chan[tn] = None
            if not self.need_dfe:
                estimate = False                                                 # Transceiver52M/Transceiver.cpp:322
            a = self.o.analyze_traffic(x, self.tsc, 3.0, req_chan=estimate)
            success = a["ok"]
            amp, toa = a["amp"], a["toa"]
            if success:
                self.energy_threshold -= 1.0
                if self.energy_threshold < 0.0:
                    self.energy_threshold = 0.0
                n2 = np.float32(np.float32(amp.imag * amp.imag) + np.float32(amp.real * amp.real))
                snr = np.float32(float(n2) / (self.energy_threshold * self.energy_threshold + 1.0))   # :340
                if estimate:
                    inv = self._inv(amp)
                    w, b = self.o.design_dfe(self.o.scale_vector(a["chan"], inv), float(snr), 7)
                    self.chan[tn] = (w, b, np.float32(a["chan_off"]))
                    self.est_time[tn] = now
            else:
                self.energy_threshold += 10.0 * math.exp(-float(fn_delta(fn, self.prev_false[0])))      # 10.0F * double -> double
                self.prev_false = now
                self.chan[tn] = None
        else:
            a = self.o.detect_rach(x, 5.0)
            success = a["ok"]
            amp, toa = a["amp"], a["toa"]
            if success:
                self.energy_threshold -= 1.0
                if self.energy_threshold < 0.0:
                    self.energy_threshold = 0.0
                self.chan[tn] = None
            else:
                self.energy_threshold += 10.0 * math.exp(-float(fn_delta(fn, self.prev_false[0])))      # 10.0F * double -> double
                self.prev_false = now
        if not success:
            return None
        if ct == RACH or not self.need_dfe:
            soft = self.o.demodulate(x, amp, toa)
        else:
            w, b, co = self.chan[tn]
            soft = self.o.equalize(self.o.scale_vector(x, self._inv(amp)), np.float32(toa - co), w, b)
        n2 = np.float32(np.float32(amp.imag * amp.imag) + np.float32(amp.real * amp.real))
        absa = np.float32(math.sqrt(float(n2)))                               # Complex::abs (Complex.h:131)
        rssi = int(math.floor(20.0 * math.log10(9450.0 / float(absa))))       # :400
        r = float(toa) * 256.0 / self.sps
        timing = int(math.floor(abs(r) + 0.5)) * (1 if r >= 0 else -1)        # round(): half away from zero (:402)
        return soft, rssi, timing

    @staticmethod
    def _inv(amp):                                                           # complex(1.0,0.0)/amp (Complex.h:85,154-160)
        n = np.float32(np.float32(amp.imag * amp.imag) + np.float32(amp.real * amp.real))
        ir, ii = np.float32(amp.real / n), np.float32(-amp.imag / n)
        one_r, one_i = np.float32(1.0), np.float32(0.0)
        return complex(np.float32(np.float32(one_r * ir) - np.float32(one_i * ii)),
                       np.float32(np.float32(one_r * ii) + np.float32(one_i * ir)))

    # ---- wire formats (:582-677) ----
    @staticmethod
    def encode_rx_datagram(tn, fn, rssi, toa, soft):
        out = bytearray(158)
        out[0] = tn & 0xff
        for i in range(4):
            out[1 + i] = (fn >> ((3 - i) * 8)) & 0xff
        out[5] = rssi & 0xff
        out[6] = (toa >> 8) & 0xff
        out[7] = toa & 0xff
        for i in range(148):
            v = float(soft[i]) * 255.0
            out[8 + i] = int(math.floor(v + 0.5)) & 0xff                     # round() of a non-negative double
        return bytes(out)

    @staticmethod
    def decode_tx_datagram(b):
        if len(b) != 154:
            return None
        sc = lambda v: v - 256 if v > 127 else v
        fn = 0
        for i in range(4):
            fn = (fn << 8) | b[1 + i]
        return sc(b[0]), fn, sc(b[5]), np.frombuffer(b[6:154], np.uint8).copy()

    # ---- transmit side (:100-181) ----
    def add_radio_vector(self, bits, rssi, tn, fn):
        m = self.o.modulate(np.asarray(bits, np.int8), 8 + (tn % 4 == 0))
        q = int(-rssi / 10)                                                  # C integer division truncates toward zero
        m = self.o.scale_vector(m, complex(np.float32(math.pow(10, q)), 0.0))
        heapq.heappush(self.queue, Queued((fn, tn), m))                      # mTransmitPriorityQueue.write (:109)

    def push_radio_vector(self, tn, fn):
        now = (fn, tn)
        while self.queue and time_less(self.queue[0].time, now):             # stale bursts go to the filler table (:142-153)
            e = heapq.heappop(self.queue)
            qfn, qtn = e.time
            self.filler[qfn % self.filler_modulus[qtn]][qtn] = e.payload
        mod = fn % self.filler_modulus[tn]
        from_queue = False
        if self.queue and self.queue[0].time == now:                         # :159-173
            self.filler[mod][tn] = heapq.heappop(self.queue).payload
            from_queue = True
        return self.filler[mod][tn], from_queue
