"""ctypes binding for oracle/libsigproc_oracle.so -- TEST INFRASTRUCTURE ONLY.

The CPU restatement of the reference sigProcLib (oracle/sigproc_oracle.c).
Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("SIGPROC_ORACLE_LIB", os.path.join(_HERE, "libsigproc_oracle.so"))   # override: sanitizer builds

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")

FULL_SPAN, OVERLAP_ONLY, START_ONLY, WITH_TAIL, NO_DELAY, CUSTOM = range(6)


class c32(C.Structure):
    _fields_ = [("r", C.c_float), ("i", C.c_float)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def c64(x):
    return np.ascontiguousarray(x, dtype=np.complex64).view(np.float32)


def _cx(z):
    z = complex(z)
    return c32(np.float32(z.real), np.float32(z.imag))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp = C.c_void_p
        L.so_ctx_size.restype = C.c_size_t
        L.so_setup.argtypes = [vp, C.c_int, C.c_int]
        for n in ("so_sinLookup", "so_cosLookup", "so_sinc"):
            getattr(L, n).argtypes = [vp, C.c_float]; getattr(L, n).restype = C.c_float
        L.so_expjLookup.argtypes = [vp, C.c_float]; L.so_expjLookup.restype = c32
        L.so_convolve.argtypes = [f32p, C.c_int, f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_uint, C.c_uint]
        L.so_correlate.argtypes = [f32p, C.c_int, f32p, C.c_int, f32p, C.c_int, C.c_int]
        L.so_scale_vector.argtypes = [f32p, C.c_int, c32, C.c_int]
        for n in ("so_dB", "so_dBinv"):
            getattr(L, n).argtypes = [C.c_float]; getattr(L, n).restype = C.c_float
        for n in ("so_vector_norm2", "so_vector_power"):
            getattr(L, n).argtypes = [f32p, C.c_int]; getattr(L, n).restype = C.c_float
        L.so_frequency_shift.argtypes = [vp, f32p, C.c_int, C.c_float, C.c_float, C.c_int, f32p]; L.so_frequency_shift.restype = C.c_float
        L.so_add_vector.argtypes = [f32p, C.c_int, f32p, C.c_int]
        L.so_mix_down.argtypes = [vp, f32p, C.c_int, C.c_longlong, C.c_float, f32p]
        L.so_offset_vector.argtypes = [f32p, C.c_int, c32, C.c_int]
        L.so_resample_vector.argtypes = [f32p, C.c_int, C.c_float, c32, f32p]
        L.so_gaussian_noise.argtypes = [C.c_int, C.c_float, c32, f32p]
        L.so_gmsk_rotate.argtypes = [vp, f32p, C.c_int, C.c_int, C.c_int]
        L.so_delay_vector.argtypes = [vp, f32p, C.c_int, C.c_float]
        L.so_interpolate_point.argtypes = [vp, f32p, C.c_int, C.c_float, C.c_int]; L.so_interpolate_point.restype = c32
        L.so_peak_detect.argtypes = [vp, f32p, C.c_int, f32p, f32p]; L.so_peak_detect.restype = c32
        L.so_modulate_gsm.argtypes = [vp, i8p, C.c_int, C.c_int, f32p]
        L.so_energy_detect.argtypes = [f32p, C.c_int, C.c_uint, C.c_float, f32p, C.c_int]
        L.so_analyze_traffic.argtypes = [vp, f32p, C.c_int, C.c_uint, C.c_float, C.c_uint, f32p, f32p, C.c_int,
                                         f32p, i32p, f32p, f32p]
        L.so_detect_rach.argtypes = [vp, f32p, C.c_int, C.c_float, f32p, f32p, f32p]
        L.so_demodulate.argtypes = [vp, f32p, C.c_int, c32, C.c_float, f32p]
        L.so_create_lpf.argtypes = [f32p, C.c_int, C.c_float, f32p]
        L.so_polyphase_resample.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p]
        L.so_design_dfe.argtypes = [vp, f32p, C.c_int, C.c_float, C.c_int, f32p, f32p]
        L.so_equalize.argtypes = [vp, f32p, C.c_int, C.c_float, f32p, C.c_int, f32p, C.c_int, f32p]
        L.so_normal_batch.argtypes = [vp, f32p, i32p, i32p, C.c_int, C.c_uint, C.c_float, u8p, f32p, f32p, f32p,
                                      C.c_int, C.c_int]
        L.so_rach_batch.argtypes = [vp, f32p, i32p, i32p, C.c_int, C.c_float, u8p, f32p, f32p, f32p,
                                    C.c_int, C.c_int]
        _lib = L
    return _lib


# so_ctx field layout (sigproc_oracle.h); offsets computed below from the same constants
_MAXSPS = 8
_TS = 1024


class Oracle:
    def __init__(self, sps, variant52m=False):
        self.L = lib()
        self.sps = sps
        self.variant52m = bool(variant52m)
        self.buf = (C.c_char * self.L.so_ctx_size())()
        self.ctx = C.cast(self.buf, C.c_void_p)
        if self.L.so_setup(self.ctx, sps, int(variant52m)) != 0:
            raise ValueError("so_setup failed")
        self._parse()

    def _parse(self):
        raw = np.frombuffer(self.buf, dtype=np.uint8)
        o = 8

        def take(n, dt):
            nonlocal o
            a = raw[o:o + n * np.dtype(dt).itemsize].view(dt).copy()
            o += n * np.dtype(dt).itemsize
            return a
        sps = self.sps
        self.cosT = take(_TS + 2, np.float32)[:_TS + 1]
        self.sinT = take(_TS + 2, np.float32)[:_TS + 1]
        self.rot = take(157 * _MAXSPS, np.complex64)[:157 * sps]
        self.rev = take(157 * _MAXSPS, np.complex64)[:157 * sps]
        plen = int(take(1, np.int32)[0])
        self.pulse = take(2 * _MAXSPS + 1, np.float32)[:plen]
        mid = take(8 * 16 * _MAXSPS, np.complex64).reshape(8, 16 * _MAXSPS)
        self.mid = mid[:, :16 * sps].copy()
        self.mid_toa = take(8, np.float32)
        self.mid_gain = take(8, np.complex64)
        self.rach = take(41 * _MAXSPS, np.complex64)[:41 * sps]
        self.rach_toa = take(1, np.float32)[0]
        self.rach_gain = take(1, np.complex64)[0]

    # scalars
    def sinc(self, x): return self.L.so_sinc(self.ctx, np.float32(x))
    def sinLookup(self, x): return self.L.so_sinLookup(self.ctx, np.float32(x))
    def cosLookup(self, x): return self.L.so_cosLookup(self.ctx, np.float32(x))

    def expjLookup(self, x):
        z = self.L.so_expjLookup(self.ctx, np.float32(x))
        return complex(z.r, z.i)

    def dB(self, x): return self.L.so_dB(np.float32(x))
    def dBinv(self, x): return self.L.so_dBinv(np.float32(x))

    # the rest of sigProcLib.h's surface
    def vector_norm2(self, x):
        x = c64(x); return np.float32(self.L.so_vector_norm2(x, x.size // 2))

    def vector_power(self, x):
        x = c64(x); return np.float32(self.L.so_vector_power(x, x.size // 2))

    def frequency_shift(self, x, freq, start_phase=0.0, real_only=False):
        x = c64(x); y = np.zeros_like(x)
        fin = self.L.so_frequency_shift(self.ctx, x, x.size // 2, np.float32(freq), np.float32(start_phase), int(real_only), y)
        return y.view(np.complex64), np.float32(fin)

    def add_vector(self, x, y):
        x = c64(x).copy(); y = c64(y)
        self.L.so_add_vector(x, x.size // 2, y, y.size // 2)
        return x.view(np.complex64)

    def mix_down(self, x, n0, freq):
        """The channeliser's mixer: frequencyShift sample by sample, the phase of raw sample n0 + k formed directly."""
        x = c64(x); y = np.zeros_like(x)
        self.L.so_mix_down(self.ctx, x, x.size // 2, int(n0), np.float32(freq), y)
        return y.view(np.complex64)

    @staticmethod
    def mix_phase(n, freq):
        t = float(n) * float(np.float32(freq))
        return np.float32(t - math.floor(t * 0.15915494309189535) * 6.283185307179586)

    def offset_vector(self, x, offset, real_only=False):
        x = c64(x).copy()
        self.L.so_offset_vector(x, x.size // 2, _cx(offset), int(real_only))
        return x.view(np.complex64)

    def resample_vector(self, x, exp_factor, end_point=0j):
        x = c64(x)
        out = np.zeros(2 * (int(np.ceil(x.size // 2 * float(exp_factor))) + 4), np.float32)
        n = self.L.so_resample_vector(x, x.size // 2, np.float32(exp_factor), _cx(end_point), out)
        return None if n < 0 else out.view(np.complex64)[:n].copy()

    def gaussian_noise(self, seed, length, variance=1.0, mean=0j):
        """srand(seed) in this process's C library, then the reference's draw order."""
        C.CDLL(None).srand(int(seed))
        out = np.zeros(2 * length, np.float32)
        self.L.so_gaussian_noise(length, np.float32(variance), _cx(mean), out)
        return out.view(np.complex64)

    # primitives
    def convolve(self, a, b, span=NO_DELAY, a_real=False, b_real=False, start=0, length=0, abssym=False):
        a = c64(a); b = c64(b)
        na, nb = a.size // 2, b.size // 2
        out = np.zeros(2 * (na + nb + 2 + length), np.float32)
        n = self.L.so_convolve(a, na, b, nb, out, span, (1 if a_real else 0) | (2 if b_real else 0) | (4 if abssym else 0), start, length)
        return None if n < 0 else out.view(np.complex64)[:n].copy()

    def correlate(self, a, b, span=NO_DELAY, a_real=False, b_real=False):
        a = c64(a); b = c64(b)
        na, nb = a.size // 2, b.size // 2
        out = np.zeros(2 * (na + nb + 2), np.float32)
        n = self.L.so_correlate(a, na, b, nb, out, span, (1 if a_real else 0) | (2 if b_real else 0))
        return None if n < 0 else out.view(np.complex64)[:n].copy()

    def delay_vector(self, x, delay):
        x = c64(x).copy()
        self.L.so_delay_vector(self.ctx, x, x.size // 2, np.float32(delay))
        return x.view(np.complex64)

    def interpolate_point(self, x, ix):
        x = c64(x)
        z = self.L.so_interpolate_point(self.ctx, x, x.size // 2, np.float32(ix), 0)
        return np.complex64(complex(z.r, z.i))

    def peak_detect(self, x):
        x = c64(x); i = np.zeros(1, np.float32); a = np.zeros(1, np.float32)
        z = self.L.so_peak_detect(self.ctx, x, x.size // 2, i, a)
        return np.complex64(complex(z.r, z.i)), i[0], a[0]

    def scale_vector(self, x, s):
        x = c64(x).copy()
        self.L.so_scale_vector(x, x.size // 2, _cx(s), 0)
        return x.view(np.complex64)

    def gmsk_rotate(self, x, reverse=False):
        x = c64(x).copy()
        self.L.so_gmsk_rotate(self.ctx, x, x.size // 2, int(reverse), 0)
        return x.view(np.complex64)

    # burst level
    def modulate(self, bits, guard):
        bits = np.ascontiguousarray(bits, np.int8)
        out = np.zeros(2 * self.sps * (bits.size + guard), np.float32)
        n = self.L.so_modulate_gsm(self.ctx, bits, bits.size, guard, out)
        return out.view(np.complex64)[:n].copy()

    def energy_detect(self, x, win, thresh):
        x = c64(x); a = np.zeros(1, np.float32)
        ok = self.L.so_energy_detect(x, x.size // 2, win, np.float32(thresh), a, int(self.variant52m))
        return bool(ok), a[0]

    def analyze_traffic(self, x, tsc, thresh=3.0, req_chan=False, max_toa=4):
        x = c64(x)
        amp = np.zeros(2, np.float32); toa = np.zeros(1, np.float32); ptm = np.zeros(1, np.float32)
        chan = np.zeros(2 * 6 * self.sps, np.float32); cl = np.zeros(1, np.int32); co = np.zeros(1, np.float32)
        ok = self.L.so_analyze_traffic(self.ctx, x, x.size // 2, tsc, np.float32(thresh), max_toa, amp, toa,
                                       int(req_chan), chan, cl, co, ptm)
        res = dict(ok=bool(ok), amp=np.complex64(complex(amp[0], amp[1])), toa=toa[0], peak_to_mean=ptm[0])
        if cl[0] > 0:
            res["chan"] = chan.view(np.complex64)[:cl[0]].copy()
            res["chan_off"] = co[0]
        return res

    def detect_rach(self, x, thresh=5.0):
        x = c64(x)
        amp = np.zeros(2, np.float32); toa = np.zeros(1, np.float32); ptm = np.zeros(1, np.float32)
        ok = self.L.so_detect_rach(self.ctx, x, x.size // 2, np.float32(thresh), amp, toa, ptm)
        return dict(ok=bool(ok), amp=np.complex64(complex(amp[0], amp[1])), toa=toa[0], peak_to_mean=ptm[0])

    def demodulate(self, x, amp, toa):
        x = c64(x)
        soft = np.zeros(x.size // 2 + 4, np.float32)
        n = self.L.so_demodulate(self.ctx, x, x.size // 2, _cx(amp), np.float32(toa), soft)
        return soft[:n].copy()

    def create_lpf(self, raw, gain):
        raw = np.ascontiguousarray(raw, np.float32)
        out = np.zeros(raw.size, np.float32)
        self.L.so_create_lpf(raw, raw.size, np.float32(gain), out)
        return out

    def polyphase_resample(self, x, P, Q, lpf):
        x = c64(x); lpf = np.ascontiguousarray(lpf, np.float32)
        n = x.size // 2
        out = np.zeros(2 * (int(np.ceil(n * P / Q)) + 4), np.float32)
        m = self.L.so_polyphase_resample(x, n, P, Q, lpf, lpf.size, out)
        return out.view(np.complex64)[:m].copy()

    def design_dfe(self, chan, snr, Nf=7):
        chan = c64(chan)
        w = np.zeros(2 * Nf, np.float32); b = np.zeros(2 * (chan.size // 2), np.float32)
        nb = self.L.so_design_dfe(self.ctx, chan, chan.size // 2, np.float32(snr), Nf, w, b)
        if nb < 0:
            return None
        return w.view(np.complex64).copy(), b.view(np.complex64)[:nb].copy()

    def equalize(self, x, toa, w, b):
        x = c64(x); w = c64(w); b = c64(b)
        soft = np.zeros(x.size // 2 + 4, np.float32)
        n = self.L.so_equalize(self.ctx, x, x.size // 2, np.float32(toa), w, w.size // 2, b, b.size // 2, soft)
        return soft[:n].copy()

    def normal_batch(self, x, off, length, tsc, thresh=3.0, nsoft=148, nthreads=1):
        x = c64(x); B = len(off)
        ok = np.zeros(B, np.uint8); amp = np.zeros(2 * B, np.float32); toa = np.zeros(B, np.float32)
        soft = np.zeros(B * nsoft, np.float32)
        self.L.so_normal_batch(self.ctx, x, np.ascontiguousarray(off, np.int32),
                               np.ascontiguousarray(length, np.int32), B, tsc, np.float32(thresh),
                               ok, amp, toa, soft, nsoft, nthreads)
        return ok, amp.view(np.complex64), toa, soft.reshape(B, nsoft)

    def rach_batch(self, x, off, length, thresh=5.0, nsoft=148, nthreads=1):
        x = c64(x); B = len(off)
        ok = np.zeros(B, np.uint8); amp = np.zeros(2 * B, np.float32); toa = np.zeros(B, np.float32)
        soft = np.zeros(B * nsoft, np.float32)
        self.L.so_rach_batch(self.ctx, x, np.ascontiguousarray(off, np.int32),
                             np.ascontiguousarray(length, np.int32), B, np.float32(thresh),
                             ok, amp, toa, soft, nsoft, nthreads)
        return ok, amp.view(np.complex64), toa, soft.reshape(B, nsoft)
