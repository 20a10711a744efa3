"""Shared helpers for the parity tests."""
import numpy as np


def beq(a, b):
    """Bit-exact equality of two arrays (dtype, shape and bytes)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def assert_beq(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, a.shape, b.shape)
    if a.tobytes() != b.tobytes():
        av = a.view(np.float32) if a.dtype == np.complex64 else a
        bv = b.view(np.float32) if b.dtype == np.complex64 else b
        bad = np.flatnonzero(av.ravel().view(np.uint8 if av.dtype.itemsize == 1 else np.uint32) !=
                             bv.ravel().view(np.uint8 if bv.dtype.itemsize == 1 else np.uint32))
        raise AssertionError("%s: %d of %d words differ, first at %d: %r vs %r" % (
            what, bad.size, av.size, bad[0], av.ravel()[bad[0]], bv.ravel()[bad[0]]))


def bursts(g):
    """Iterate (index, samples) over a packed golden file."""
    for i, (o, n) in enumerate(zip(g["off"], g["len"])):
        yield i, g["x"][o:o + n]
