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


def assert_veq(a, b, what=""):
    """Value-exact equality (IEEE ==, so -0 == +0); no NaNs expected."""
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if not np.array_equal(a, b):
        bad = np.flatnonzero(~(a.ravel() == b.ravel()))
        raise AssertionError("%s: %d of %d values differ, first at %d: %r vs %r" % (
            what, bad.size, a.size, bad[0], a.ravel()[bad[0]], b.ravel()[bad[0]]))


class GpuBatch:
    """A packed batch of bursts resident on cuda:0 plus output buffers (torch is plumbing only)."""

    def __init__(self, x, off, length, nsoft=148, stride=None, device="cuda:0"):
        import torch
        self.torch = torch
        self.B = len(off)
        self.nsoft = nsoft
        self.stride = stride or nsoft
        dev = torch.device(device)
        self.x = torch.from_numpy(np.ascontiguousarray(x, np.complex64).view(np.float32)).to(dev)
        self.off = torch.from_numpy(np.ascontiguousarray(off, np.int32)).to(dev)
        self.len = torch.from_numpy(np.ascontiguousarray(length, np.int32)).to(dev)
        self.flags = torch.zeros(self.B, dtype=torch.uint8, device=dev)
        self.amp = torch.zeros(self.B, 2, dtype=torch.float32, device=dev)
        self.toa = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self.pwr = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self.soft = torch.full((self.B, self.stride), -1.0, dtype=torch.float32, device=dev)
        self.hard = torch.full((self.B, self.stride), 255, dtype=torch.uint8, device=dev)

    def results(self):
        self.torch.cuda.synchronize()
        return dict(flags=self.flags.cpu().numpy(), amp=self.amp.cpu().numpy().view(np.complex64).ravel(),
                    toa=self.toa.cpu().numpy(), pwr=self.pwr.cpu().numpy(), soft=self.soft.cpu().numpy(),
                    hard=self.hard.cpu().numpy())
