"""ctypes binding for oracle/libfec_oracle.so (the CPU restatement of the L1 FEC soft decode) --
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
XCCH_POLY, RACH_POLY = 0x10004820009, 0x06f


class FecOracle:
    def __init__(self):
        self.lib = L = C.CDLL(os.environ.get("FEC_ORACLE_LIB", os.path.join(_HERE, "libfec_oracle.so")))   # override: sanitizer builds
        L.fo_encode.argtypes = [u8p, C.c_int, u8p]
        L.fo_viterbi_decode.argtypes = [f32p, C.c_int, u8p, C.c_int]
        for n in ("fo_parity", "fo_syndrome"):
            getattr(L, n).argtypes = [C.c_uint64, C.c_uint, u8p, C.c_int]
            getattr(L, n).restype = C.c_uint64
        L.fo_lsb8msb.argtypes = [u8p, C.c_int]
        L.fo_wire.argtypes = [C.c_float]; L.fo_wire.restype = C.c_float
        L.fo_xcch_decode.argtypes = [f32p, u8p, u8p, C.POINTER(C.c_uint64)]
        L.fo_rach_decode.argtypes = [f32p, u8p, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
        L.fo_xcch_encode.argtypes = [u8p, u8p, u8p]
        L.fo_tch_decode.argtypes = [f32p, u8p, u8p]
        L.fo_tch_decode_batch.argtypes = [f32p, C.c_int, C.c_int, C.c_int, u8p, u8p, u8p, u8p, u8p, C.c_int]
        L.fo_xcch_decode_batch.argtypes = [f32p, C.c_int, C.c_int, C.c_int, u8p, u8p, C.c_int]
        L.fo_rach_decode_batch.argtypes = [f32p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]

    def encode(self, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        out = np.zeros(2 * len(bits), np.uint8)
        self.lib.fo_encode(bits, len(bits), out)
        return out

    def viterbi_decode(self, soft, nout):
        soft = np.ascontiguousarray(soft, np.float32)
        out = np.zeros(nout, np.uint8)
        self.lib.fo_viterbi_decode(soft, len(soft), out, nout)
        return out

    def parity(self, coeff, psize, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        return int(self.lib.fo_parity(coeff, psize, bits, len(bits)))

    def syndrome(self, coeff, psize, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        return int(self.lib.fo_syndrome(coeff, psize, bits, len(bits)))

    def lsb8msb(self, bits):
        b = np.ascontiguousarray(bits, np.uint8).copy()
        self.lib.fo_lsb8msb(b, len(b))
        return b

    def wire(self, v):
        return np.array([self.lib.fo_wire(float(x)) for x in np.asarray(v, np.float32).ravel()], np.float32).reshape(np.shape(v))

    def xcch_decode(self, i4x114):
        u = np.zeros(228, np.uint8); d = np.zeros(184, np.uint8); syn = C.c_uint64()
        ok = self.lib.fo_xcch_decode(np.ascontiguousarray(i4x114, np.float32).ravel(), u, d, C.byref(syn))
        return dict(ok=bool(ok), u=u, d=d, syndrome=int(syn.value))

    def rach_decode(self, e36):
        u = np.zeros(18, np.uint8); bsic = C.c_uint(); ra = C.c_uint()
        t = self.lib.fo_rach_decode(np.ascontiguousarray(e36, np.float32), u, C.byref(bsic), C.byref(ra))
        return dict(tail_ok=bool(t), u=u, bsic=int(bsic.value), ra=int(ra.value))

    def xcch_encode(self, frame23, tsc26):
        out = np.zeros(4 * 148, np.uint8)
        self.lib.fo_xcch_encode(np.ascontiguousarray(frame23, np.uint8), np.ascontiguousarray(tsc26, np.uint8), out)
        return out.reshape(4, 148)

    def tch_decode(self, c456):
        u = np.zeros(189, np.uint8); d = np.zeros(260, np.uint8)
        good = self.lib.fo_tch_decode(np.ascontiguousarray(c456, np.float32), u, d)
        return dict(good=bool(good), u=u, d=d)

    def tch_decode_batch(self, soft, wire=True, nthreads=8):
        soft = np.ascontiguousarray(soft, np.float32)
        nblk = soft.shape[0] // 4 - 1
        tch = np.zeros((nblk, 33), np.uint8); good = np.zeros(nblk, np.uint8)
        facch = np.zeros((nblk, 23), np.uint8); fok = np.zeros(nblk, np.uint8); stolen = np.zeros(nblk, np.uint8)
        self.lib.fo_tch_decode_batch(soft, soft.shape[1], soft.shape[0], int(wire), tch, good, facch, fok, stolen, nthreads)
        return dict(tch=tch, good=good, facch=facch, facch_ok=fok, stolen=stolen)

    def xcch_decode_batch(self, soft, wire=True, nthreads=8):
        soft = np.ascontiguousarray(soft, np.float32)
        nblk = soft.shape[0] // 4
        frames = np.zeros((nblk, 23), np.uint8); ok = np.zeros(nblk, np.uint8)
        self.lib.fo_xcch_decode_batch(soft, soft.shape[1], nblk, int(wire), frames, ok, nthreads)
        return frames, ok

    def rach_decode_batch(self, soft, wire=True, nthreads=8):
        soft = np.ascontiguousarray(soft, np.float32)
        out = np.zeros((soft.shape[0], 3), np.uint8)
        self.lib.fo_rach_decode_batch(soft, soft.shape[1], soft.shape[0], int(wire), out, nthreads)
        return out
