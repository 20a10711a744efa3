"""ctypes binding for oracle/_ref/libref_fec.so (the REAL reference BitVector / ViterbiR2O4 / Parity code,
compiled in place by `make -C oracle ref`) -- TEST INFRASTRUCTURE ONLY.  Used to pin oracle/fec_oracle.c and
to generate tests/golden/fec_*.npz."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_fec.so")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def available():
    return os.path.exists(_PATH)


class RefFec:
    def __init__(self):
        self.lib = L = C.CDLL(_PATH)
        L.reffec_soft_decode.argtypes = [f32p, C.c_int, u8p, C.c_int]
        L.reffec_encode.argtypes = [u8p, C.c_int, u8p]
        for n in ("reffec_parity", "reffec_syndrome"):
            getattr(L, n).argtypes = [C.c_uint64, C.c_uint, C.c_uint, u8p, C.c_int]
            getattr(L, n).restype = C.c_uint64
        L.reffec_lsb8msb.argtypes = [u8p, C.c_int]
        L.reffec_xcch_encode.argtypes = [u8p, u8p]
        L.reffec_xcch_decode.argtypes = [f32p, u8p, u8p, C.POINTER(C.c_uint64)]
        L.reffec_rach_decode.argtypes = [f32p, u8p, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
        L.reffec_tch_encode.argtypes = [u8p, u8p]
        L.reffec_tch_decode.argtypes = [f32p, u8p, u8p]

    def tch_encode(self, d260):
        out = np.zeros(456, np.uint8)
        self.lib.reffec_tch_encode(np.ascontiguousarray(d260, np.uint8), out)
        return out

    def tch_decode(self, c456):
        u = np.zeros(189, np.uint8); d = np.zeros(260, np.uint8)
        good = self.lib.reffec_tch_decode(np.ascontiguousarray(c456, np.float32), u, d)
        return dict(good=bool(good), u=u, d=d)

    def soft_decode(self, soft, nout):
        soft = np.ascontiguousarray(soft, np.float32)
        out = np.zeros(nout, np.uint8)
        self.lib.reffec_soft_decode(soft, len(soft), out, nout)
        return out

    def encode(self, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        out = np.zeros(2 * len(bits), np.uint8)
        self.lib.reffec_encode(bits, len(bits), out)
        return out

    def parity(self, coeff, psize, cwsize, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        return int(self.lib.reffec_parity(coeff, psize, cwsize, bits, len(bits)))

    def syndrome(self, coeff, psize, cwsize, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        return int(self.lib.reffec_syndrome(coeff, psize, cwsize, bits, len(bits)))

    def lsb8msb(self, bits):
        b = np.ascontiguousarray(bits, np.uint8).copy()
        self.lib.reffec_lsb8msb(b, len(b))
        return b

    def xcch_encode(self, d184):
        out = np.zeros(4 * 114, np.uint8)
        self.lib.reffec_xcch_encode(np.ascontiguousarray(d184, np.uint8), out)
        return out.reshape(4, 114)

    def xcch_decode(self, i4x114):
        u = np.zeros(228, np.uint8); d = np.zeros(184, np.uint8); syn = C.c_uint64()
        ok = self.lib.reffec_xcch_decode(np.ascontiguousarray(i4x114, np.float32).ravel(), u, d, C.byref(syn))
        return dict(ok=bool(ok), u=u, d=d, syndrome=int(syn.value))

    def rach_decode(self, e36):
        u = np.zeros(18, np.uint8); bsic = C.c_uint(); ra = C.c_uint()
        tail_ok = self.lib.reffec_rach_decode(np.ascontiguousarray(e36, np.float32), u, C.byref(bsic), C.byref(ra))
        return dict(tail_ok=bool(tail_ok), u=u, bsic=int(bsic.value), ra=int(ra.value))
