"""ctypes binding for oracle/_ref/libref_sigproc*.so -- TEST INFRASTRUCTURE ONLY.

The library is the *real* reference sigProcLib compiled in place from
/root/reference by `make -C oracle ref` (build container only).  It is used to
validate the CPU restatement (oracle/sigproc_oracle.c) and to generate the
committed golden fixtures (oracle/gen_golden.py).  Nothing in the product path
imports this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")

FULL_SPAN, OVERLAP_ONLY, START_ONLY, WITH_TAIL, NO_DELAY = range(5)


def available(variant=""):
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_sigproc%s.so" % variant))


def c64(x):
    """complex64 array -> float32 view (interleaved re,im), contiguous."""
    return np.ascontiguousarray(x, dtype=np.complex64).view(np.float32)


def gsm_time(lib, op, a, b=(0, 0), step=0):
    """GSM::Time of the compiled reference: op 0 a<b, 1 a>b, 2 a==b, 3 a-b, 4 FNDelta(a.fn, b.fn) -> int;
    op 5 incTN(step), 6 decTN(step), 7 a += step, 8 a + b -> (fn, tn)."""
    out, ofn, otn = C.c_int(), C.c_int(), C.c_int()
    rc = lib.ref_gsm_time(op, int(a[0]), int(a[1]), int(b[0]), int(b[1]), int(step), C.byref(out), C.byref(ofn), C.byref(otn))
    assert rc == 0
    return out.value if op <= 4 else (ofn.value, otn.value)


class Ref:
    """One loaded reference library (variant '' = Transceiver/, '52m' = Transceiver52M/).

    The reference keeps its state in process globals, so one instance == one sps.
    """

    def __init__(self, sps, variant=""):
        path = os.path.join(_HERE, "_ref", "libref_sigproc%s.so" % variant)
        # RTLD_LOCAL + a private copy per variant keeps the two variants' globals apart
        self.lib = L = C.CDLL(path, mode=C.RTLD_LOCAL)
        self.variant = variant
        self.sps = sps
        L.ref_setup.argtypes = [C.c_int]
        L.ref_get_trig_tables.argtypes = [f32p, f32p]
        L.ref_get_rotation.argtypes = [f32p, f32p]
        L.ref_get_pulse.argtypes = [f32p]
        L.ref_get_midamble.argtypes = [C.c_int, f32p, f32p, f32p]
        L.ref_get_rach.argtypes = [f32p, f32p, f32p]
        L.ref_get_gsm_bits.argtypes = [i8p, i8p, i8p]
        L.ref_get_lpf_raw.argtypes = [f32p, f32p]
        for n in ("ref_sinc", "ref_sinLookup", "ref_cosLookup"):
            getattr(L, n).argtypes = [C.c_float]
            getattr(L, n).restype = C.c_float
        L.ref_expjLookup.argtypes = [C.c_float, f32p]
        if hasattr(L, "ref_gsm_time"):
            L.ref_gsm_time.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_int)] * 3
        L.ref_convolve.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_int, f32p]
        L.ref_correlate.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_int, f32p]
        L.ref_delay_vector.argtypes = [f32p, C.c_int, C.c_float]
        L.ref_interpolate_point.argtypes = [f32p, C.c_int, C.c_float, f32p]
        L.ref_peak_detect.argtypes = [f32p, C.c_int, f32p, f32p, f32p]
        L.ref_scale_vector.argtypes = [f32p, C.c_int, C.c_float, C.c_float]
        L.ref_gmsk_rotate.argtypes = [f32p, C.c_int, C.c_int]
        L.ref_modulate.argtypes = [i8p, C.c_int, C.c_int, f32p]
        L.ref_energy_detect.argtypes = [f32p, C.c_int, C.c_uint, C.c_float, f32p]
        L.ref_analyze_traffic.argtypes = [f32p, C.c_int, C.c_uint, C.c_float, C.c_int, f32p, f32p,
                                          C.c_int, f32p, i32p, f32p]
        L.ref_detect_rach.argtypes = [f32p, C.c_int, C.c_float, f32p, f32p]
        L.ref_demodulate.argtypes = [f32p, C.c_int, C.c_float, C.c_float, C.c_float, f32p]
        L.ref_create_lpf651.argtypes = [C.c_float, f32p]
        L.ref_polyphase_resample.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p]
        L.ref_design_dfe.argtypes = [f32p, C.c_int, C.c_float, C.c_int, f32p, f32p]
        L.ref_equalize.argtypes = [f32p, C.c_int, C.c_float, f32p, C.c_int, f32p, C.c_int, f32p]
        L.ref_normal_batch.argtypes = [f32p, i32p, i32p, C.c_int, C.c_uint, C.c_float, u8p, f32p, f32p, f32p]
        L.ref_rach_batch.argtypes = [f32p, i32p, i32p, C.c_int, C.c_float, u8p, f32p, f32p, f32p]
        if hasattr(L, "ref_eq_batch"):
            L.ref_eq_batch.argtypes = [f32p, i32p, i32p, C.c_int, C.c_uint, C.c_float, C.c_float, C.c_int, u8p, f32p]
        if hasattr(L, "ref_dB"):                              # the rest of sigProcLib.h's surface
            for n in ("ref_dB", "ref_dBinv"):
                getattr(L, n).argtypes = [C.c_float]; getattr(L, n).restype = C.c_float
            for n in ("ref_vector_norm2", "ref_vector_power"):
                getattr(L, n).argtypes = [f32p, C.c_int]; getattr(L, n).restype = C.c_float
            L.ref_frequency_shift.argtypes = [f32p, C.c_int, C.c_float, C.c_float, C.c_int, f32p]; L.ref_frequency_shift.restype = C.c_float
            L.ref_add_vector.argtypes = [f32p, C.c_int, f32p, C.c_int]
            L.ref_offset_vector.argtypes = [f32p, C.c_int, C.c_float, C.c_float, C.c_int]
            L.ref_resample_vector.argtypes = [f32p, C.c_int, C.c_float, C.c_float, C.c_float, f32p]
            L.ref_gaussian_noise.argtypes = [C.c_uint, C.c_int, C.c_float, C.c_float, C.c_float, f32p]
        if L.ref_setup(sps) != 0:
            raise RuntimeError("ref_setup failed")

    # ---- tables ----
    def trig_tables(self):
        c = np.zeros(1025, np.float32); s = np.zeros(1025, np.float32)
        self.lib.ref_get_trig_tables(c, s)
        return c, s

    def rotation(self):
        n = 157 * self.sps
        a = np.zeros(2 * n, np.float32); b = np.zeros(2 * n, np.float32)
        self.lib.ref_get_rotation(a, b)
        return a.view(np.complex64), b.view(np.complex64)

    def pulse(self):
        a = np.zeros(2 * (2 * self.sps + 1), np.float32)
        n = self.lib.ref_get_pulse(a)
        return a.view(np.complex64)[:n].copy()

    def midamble(self, tsc):
        a = np.zeros(2 * 16 * self.sps, np.float32)
        toa = np.zeros(1, np.float32); g = np.zeros(2, np.float32)
        n = self.lib.ref_get_midamble(tsc, a, toa, g)
        return a.view(np.complex64)[:n].copy(), float(toa[0]), complex(g[0], g[1])

    def rach(self):
        a = np.zeros(2 * 41 * self.sps, np.float32)
        toa = np.zeros(1, np.float32); g = np.zeros(2, np.float32)
        n = self.lib.ref_get_rach(a, toa, g)
        return a.view(np.complex64)[:n].copy(), float(toa[0]), complex(g[0], g[1])

    def gsm_bits(self):
        t = np.zeros(8 * 26, np.int8); d = np.zeros(148, np.int8); r = np.zeros(41, np.int8)
        self.lib.ref_get_gsm_bits(t, d, r)
        return t.reshape(8, 26), d, r

    def lpf_raw(self):
        a = np.zeros(651, np.float32); b = np.zeros(960, np.float32)
        self.lib.ref_get_lpf_raw(a, b)
        return a, b

    # ---- scalars ----
    def sinc(self, x): return self.lib.ref_sinc(np.float32(x))
    def sinLookup(self, x): return self.lib.ref_sinLookup(np.float32(x))
    def cosLookup(self, x): return self.lib.ref_cosLookup(np.float32(x))

    def expjLookup(self, x):
        o = np.zeros(2, np.float32)
        self.lib.ref_expjLookup(np.float32(x), o)
        return complex(o[0], o[1])

    def dB(self, x): return self.lib.ref_dB(np.float32(x))
    def dBinv(self, x): return self.lib.ref_dBinv(np.float32(x))

    # ---- the rest of sigProcLib.h's surface ----
    def vector_norm2(self, x):
        x = c64(x); return np.float32(self.lib.ref_vector_norm2(x, x.size // 2))

    def vector_power(self, x):
        x = c64(x); return np.float32(self.lib.ref_vector_power(x, x.size // 2))

    def frequency_shift(self, x, freq, start_phase=0.0, real_only=False):
        x = c64(x); y = np.zeros_like(x)
        fin = self.lib.ref_frequency_shift(x, x.size // 2, np.float32(freq), np.float32(start_phase), int(real_only), y)
        return y.view(np.complex64), np.float32(fin)

    def add_vector(self, x, y):
        x = c64(x).copy(); y = c64(y)
        self.lib.ref_add_vector(x, x.size // 2, y, y.size // 2)
        return x.view(np.complex64)

    def offset_vector(self, x, offset, real_only=False):
        x = c64(x).copy(); o = complex(offset)
        self.lib.ref_offset_vector(x, x.size // 2, np.float32(o.real), np.float32(o.imag), int(real_only))
        return x.view(np.complex64)

    def resample_vector(self, x, exp_factor, end_point=0j):
        x = c64(x); e = complex(end_point)
        out = np.zeros(2 * (int(np.ceil(x.size // 2 * float(exp_factor))) + 4), np.float32)
        n = self.lib.ref_resample_vector(x, x.size // 2, np.float32(exp_factor), np.float32(e.real), np.float32(e.imag), out)
        return None if n < 0 else out.view(np.complex64)[:n].copy()

    def gaussian_noise(self, seed, length, variance=1.0, mean=0j):
        m = complex(mean); out = np.zeros(2 * length, np.float32)
        self.lib.ref_gaussian_noise(int(seed), length, np.float32(variance), np.float32(m.real), np.float32(m.imag), out)
        return out.view(np.complex64)

    # ---- primitives ----
    def _conv(self, fn, a, b, span, a_real, b_real, abssym=False):
        a = c64(a); b = c64(b)
        na, nb = a.size // 2, b.size // 2
        out = np.zeros(2 * (na + nb + 2), np.float32)
        n = fn(a, na, b, nb, span, (1 if a_real else 0) | (2 if b_real else 0) | (4 if abssym else 0), out)
        if n < 0:
            return None
        return out.view(np.complex64)[:n].copy()

    def convolve(self, a, b, span=NO_DELAY, a_real=False, b_real=False, abssym=False):
        return self._conv(self.lib.ref_convolve, a, b, span, a_real, b_real, abssym)

    def correlate(self, a, b, span=NO_DELAY, a_real=False, b_real=False):
        return self._conv(self.lib.ref_correlate, a, b, span, a_real, b_real)

    def delay_vector(self, x, delay):
        x = c64(x).copy()
        self.lib.ref_delay_vector(x, x.size // 2, delay)
        return x.view(np.complex64)

    def interpolate_point(self, x, ix):
        x = c64(x); o = np.zeros(2, np.float32)
        self.lib.ref_interpolate_point(x, x.size // 2, ix, o)
        return np.complex64(complex(o[0], o[1]))

    def peak_detect(self, x):
        x = c64(x); p = np.zeros(2, np.float32); i = np.zeros(1, np.float32); a = np.zeros(1, np.float32)
        self.lib.ref_peak_detect(x, x.size // 2, p, i, a)
        return np.complex64(complex(p[0], p[1])), i[0], a[0]

    def scale_vector(self, x, s):
        x = c64(x).copy()
        self.lib.ref_scale_vector(x, x.size // 2, np.float32(s.real), np.float32(s.imag))
        return x.view(np.complex64)

    def gmsk_rotate(self, x, reverse=False):
        x = c64(x).copy()
        self.lib.ref_gmsk_rotate(x, x.size // 2, int(reverse))
        return x.view(np.complex64)

    # ---- burst level ----
    def modulate(self, bits, guard):
        bits = np.ascontiguousarray(bits, np.int8)
        out = np.zeros(2 * self.sps * (bits.size + guard), np.float32)
        n = self.lib.ref_modulate(bits, bits.size, guard, out)
        return out.view(np.complex64)[:n].copy()

    def energy_detect(self, x, win, thresh):
        x = c64(x); a = np.zeros(1, np.float32)
        ok = self.lib.ref_energy_detect(x, x.size // 2, win, thresh, a)
        return bool(ok), a[0]

    def analyze_traffic(self, x, tsc, thresh=3.0, req_chan=False, max_toa=4):
        x = c64(x)
        amp = np.zeros(2, np.float32); toa = np.zeros(1, np.float32)
        chan = np.zeros(2 * 6 * self.sps, np.float32); cl = np.zeros(1, np.int32); co = np.zeros(1, np.float32)
        ok = self.lib.ref_analyze_traffic(x, x.size // 2, tsc, thresh, max_toa, amp, toa,
                                          int(req_chan), chan, cl, co)
        res = dict(ok=bool(ok), amp=np.complex64(complex(amp[0], amp[1])), toa=toa[0])
        if cl[0] > 0:
            res["chan"] = chan.view(np.complex64)[:cl[0]].copy()
            res["chan_off"] = co[0]
        return res

    def detect_rach(self, x, thresh=5.0):
        x = c64(x)
        amp = np.zeros(2, np.float32); toa = np.zeros(1, np.float32)
        ok = self.lib.ref_detect_rach(x, x.size // 2, thresh, amp, toa)
        return dict(ok=bool(ok), amp=np.complex64(complex(amp[0], amp[1])), toa=toa[0])

    def demodulate(self, x, amp, toa):
        x = c64(x)
        soft = np.zeros(x.size // 2 + 4, np.float32)
        n = self.lib.ref_demodulate(x, x.size // 2, np.float32(amp.real), np.float32(amp.imag),
                                    np.float32(toa), soft)
        return soft[:n].copy()

    def create_lpf651(self, gain):
        o = np.zeros(651, np.float32)
        self.lib.ref_create_lpf651(gain, o)
        return o

    def polyphase_resample(self, x, P, Q, lpf):
        x = c64(x); lpf = np.ascontiguousarray(lpf, np.float32)
        n = x.size // 2
        out = np.zeros(2 * (int(np.ceil(n * P / Q)) + 4), np.float32)
        m = self.lib.ref_polyphase_resample(x, n, P, Q, lpf, lpf.size, out)
        return out.view(np.complex64)[:m].copy()

    def design_dfe(self, chan, snr, Nf=7):
        chan = c64(chan)
        w = np.zeros(2 * Nf, np.float32); b = np.zeros(2 * (chan.size // 2), np.float32)
        nb = self.lib.ref_design_dfe(chan, chan.size // 2, snr, Nf, w, b)
        if nb < 0:
            return None
        return w.view(np.complex64).copy(), b.view(np.complex64)[:nb].copy()

    def equalize(self, x, toa, w, b):
        x = c64(x); w = c64(w); b = c64(b)
        soft = np.zeros(x.size // 2 + 4, np.float32)
        n = self.lib.ref_equalize(x, x.size // 2, np.float32(toa), w, w.size // 2, b, b.size // 2, soft)
        return soft[:n].copy()

    def normal_batch(self, x, off, length, tsc, thresh=3.0):
        x = c64(x); B = len(off)
        ok = np.zeros(B, np.uint8); amp = np.zeros(2 * B, np.float32); toa = np.zeros(B, np.float32)
        soft = np.zeros(B * 148, np.float32)
        self.lib.ref_normal_batch(x, np.ascontiguousarray(off, np.int32), np.ascontiguousarray(length, np.int32),
                                  B, tsc, thresh, ok, amp, toa, soft)
        return ok, amp.view(np.complex64), toa, soft.reshape(B, 148)

    def eq_batch(self, x, off, length, tsc, detect_thresh=3.0, energy_thresh=10.0, max_toa=4):
        """ref_eq_batch: the equalised receive leg burst by burst inside the reference (bench.py's config-5 baseline)."""
        x = c64(x); B = len(off)
        ok = np.zeros(B, np.uint8); soft = np.zeros(B * 157, np.float32)
        self.lib.ref_eq_batch(x, np.ascontiguousarray(off, np.int32), np.ascontiguousarray(length, np.int32), B, tsc,
                              np.float32(detect_thresh), np.float32(energy_thresh), max_toa, ok, soft)
        return ok, soft.reshape(B, 157)

    def rach_batch(self, x, off, length, thresh=5.0):
        x = c64(x); B = len(off)
        ok = np.zeros(B, np.uint8); amp = np.zeros(2 * B, np.float32); toa = np.zeros(B, np.float32)
        soft = np.zeros(B * 148, np.float32)
        self.lib.ref_rach_batch(x, np.ascontiguousarray(off, np.int32), np.ascontiguousarray(length, np.int32),
                                B, thresh, ok, amp, toa, soft)
        return ok, amp.view(np.complex64), toa, soft.reshape(B, 148)
