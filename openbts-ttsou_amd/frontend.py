"""ctypes view of include/trxsig_frontend.h for tests/ and bench.py: the RadioInterface's receive front end
(pullBuffer + driveReceiveRadio, Transceiver/radioInterface.cpp:197-273, 359-401) and transmit back end
(driveTransmitRadio + pushBuffer, :123-194, 337-357) for S independent ARFCN streams (BASELINE config 4 and the TX chain).
Everything -- buffers, history, 157/156/156/156 slicing, chunking, the fused convert + resample kernels -- lives in
libtrxsig (csrc/trxsig_frontend.cpp, csrc/trxsig_tx.hip); this module only marshals pointers (torch is plumbing)."""
import ctypes as C

import numpy as np

OUTRATE = 96
OUTCHUNK = 9 * OUTRATE          # 864
OUTHISTORY = 2 * OUTRATE        # 192


class _DevView:
    """A device buffer owned by the library, presented to torch (zero copy) through __cuda_array_interface__."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(data=(int(ptr), False), shape=tuple(shape), typestr=typestr, version=2)


def _bind(L):
    vp, i32 = C.c_void_p, C.c_int
    if getattr(L, "_frontend_bound", False):
        return
    L.trxsig_rxfe_create.argtypes = [C.POINTER(vp), vp, i32, i32, vp, i32, i32, i32]
    L.trxsig_rxfe_destroy.argtypes = [vp]; L.trxsig_rxfe_destroy.restype = None
    L.trxsig_rxfe_push.argtypes = [vp, vp, i32]
    L.trxsig_rxfe_pop.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, i32, C.POINTER(i32)]
    L.trxsig_rxfe_pending.argtypes = [vp]
    L.trxsig_rxfe_push_detect_demod_normal.argtypes = [vp, vp, i32, i32, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp, i32, i32, vp, i32,
                                                       C.POINTER(i32)]
    L.trxsig_rxfe_create_wideband.argtypes = [C.POINTER(vp), vp, i32, i32, vp, i32, i32, vp, i32, i32, i32]
    L.trxsig_rxfe_push_wideband.argtypes = [vp, vp, i32]
    L.trxsig_rxfe_set_shared_filter.argtypes = [vp, i32]
    L.trxsig_txbe_create.argtypes = [C.POINTER(vp), vp, i32, i32, vp, i32, C.c_float]
    L.trxsig_txbe_destroy.argtypes = [vp]; L.trxsig_txbe_destroy.restype = None
    L.trxsig_txbe_push_bursts.argtypes = [vp, vp, vp, vp, i32]
    L.trxsig_txbe_pop.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(i32)]
    L.trxsig_txbe_pending.argtypes = [vp]
    L.trxsig_txbe_set_fused.argtypes = [vp, i32]
    L._frontend_bound = True


class RxFrontEnd:
    def __init__(self, ctx, n_streams, lpf_taps, device="cuda:0", swap_iq=True, max_chunks=1, start_tn=0, carrier_freq=None, rate_factor=0):
        """carrier_freq (radians per wideband sample, one per carrier) + rate_factor: the channeliser -- n_streams WIDEBAND
        streams at rate_factor x 400 kS/s, len(carrier_freq) ARFCNs each; bursts come out per (stream, carrier)."""
        import torch
        self.torch = torch
        self.ctx = ctx
        self.L = ctx.L
        _bind(self.L)
        self.sps = ctx.sps
        self.dev = torch.device(device)
        lpf = np.ascontiguousarray(lpf_taps, np.float32)
        h = C.c_void_p()
        self.rate_factor = rate_factor
        if carrier_freq is not None:
            fr = np.ascontiguousarray(carrier_freq, np.float32)
            ctx._chk(self.L.trxsig_rxfe_create_wideband(C.byref(h), ctx.h, n_streams, fr.size, fr.ctypes.data, rate_factor, max_chunks,
                                                        lpf.ctypes.data, lpf.size, int(swap_iq), start_tn), "trxsig_rxfe_create_wideband")
            self.S = n_streams * fr.size
            self.Sw = n_streams
        else:
            ctx._chk(self.L.trxsig_rxfe_create(C.byref(h), ctx.h, n_streams, max_chunks, lpf.ctypes.data, lpf.size, int(swap_iq),
                                               start_tn), "trxsig_rxfe_create")
            self.S = n_streams
        self.h = h

    def set_shared_filter(self, on=True):
        """The channeliser's shared-filter form (carriers on the grid of sixteenths of the wideband rate): one pass over the raw
        samples for all carriers, ~1e-6 from the per-carrier form instead of bit-equal."""
        self.ctx._chk(self.L.trxsig_rxfe_set_shared_filter(self.h, int(on)), "trxsig_rxfe_set_shared_filter")

    def push_wideband(self, iq):
        """iq: int16 tensor [Sw, K*864*rate_factor, 2] (device), K whole chunks per wideband stream."""
        torch = self.torch
        n = OUTCHUNK * self.rate_factor
        assert iq.dtype == torch.int16 and iq.shape[0] == self.Sw and iq.shape[1] % n == 0 and iq.shape[2] == 2
        iq = iq.contiguous()
        self.ctx._chk(self.L.trxsig_rxfe_push_wideband(self.h, iq.data_ptr(), iq.shape[1] // n), "trxsig_rxfe_push_wideband")
        self._keep = iq

    def close(self):
        if self.h:
            self.L.trxsig_rxfe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push_chunk(self, iq):
        """iq: int16 tensor [S, K*864, 2] (device), K whole chunks per stream."""
        torch = self.torch
        assert iq.dtype == torch.int16 and iq.shape[0] == self.S and iq.shape[1] % OUTCHUNK == 0 and iq.shape[2] == 2
        iq = iq.contiguous()
        self.ctx._chk(self.L.trxsig_rxfe_push(self.h, iq.data_ptr(), iq.shape[1] // OUTCHUNK), "trxsig_rxfe_push")
        self._keep = iq                                       # the launch reads it asynchronously

    def pop_raw(self, max_bursts=4096):
        """(samples ptr, offset ptr, length ptr, tn int32 [nb], nb) -- device addresses owned by the library -- or None."""
        ps, po, pl = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nb = C.c_int()
        tn = np.zeros(max_bursts, np.int32)
        self.ctx._chk(self.L.trxsig_rxfe_pop(self.h, C.byref(ps), C.byref(po), C.byref(pl), tn.ctypes.data, max_bursts, C.byref(nb)),
                      "trxsig_rxfe_pop")
        if nb.value == 0:
            return None
        return ps.value, po.value, pl.value, tn[:nb.value].copy(), nb.value

    def pop_bursts(self):
        """Cut every stream's buffer into bursts.  Returns (samples float32 [total, 2] view of the library's receive
        buffers, offset int32 [S*nb], length int32 [S*nb], tn int32 [S*nb]) with bursts ordered stream-major, or None when
        less than one burst is buffered.  The tensors alias library memory: valid until the next push."""
        r = self.pop_raw()
        if r is None:
            return None
        ps, po, pl, tn, nb = r
        torch = self.torch
        B = self.S * nb
        off = torch.as_tensor(_DevView(po, (B,), "<i4"), device=self.dev)
        length = torch.as_tensor(_DevView(pl, (B,), "<i4"), device=self.dev)
        torch.cuda.synchronize()
        end = int((off.to(torch.int64) + length.to(torch.int64)).max().item())
        x = torch.as_tensor(_DevView(ps, (end, 2), "<f4"), device=self.dev)
        return x, off, length, np.tile(tn, self.S)

    def push_detect_demod(self, iq, tsc, flags, amp, toa, soft, avgpwr=None, hard=None, detect_thresh=3.0, energy_thresh=0.0, nsoft=148,
                          soft_stride=None, max_bursts=4096):
        """The fused call: iq int16 [S, K*864, 2] (device); outputs are device tensors with room for S * bursts entries
        (burst j of stream s at s*nb + j).  Returns (nb bursts per stream, tn int32 [nb])."""
        torch = self.torch
        assert iq.dtype == torch.int16 and iq.shape[0] == self.S and iq.shape[1] % OUTCHUNK == 0 and iq.shape[2] == 2
        iq = iq.contiguous()
        if soft_stride is None:
            soft_stride = soft.shape[-1]
        nb = C.c_int()
        max_bursts = min(max_bursts, flags.numel() // self.S)    # cap_tn: what the output arrays hold, in bursts per stream
        tn = np.zeros(max(max_bursts, 1), np.int32)
        p = lambda t: None if t is None else t.data_ptr()
        self.ctx._chk(self.L.trxsig_rxfe_push_detect_demod_normal(self.h, iq.data_ptr(), iq.shape[1] // OUTCHUNK, tsc, detect_thresh,
                                                                  energy_thresh, p(flags), p(amp), p(toa), p(avgpwr), p(soft), p(hard),
                                                                  nsoft, soft_stride, tn.ctypes.data, max_bursts, C.byref(nb)),
                      "trxsig_rxfe_push_detect_demod_normal")
        self._keep = iq
        return nb.value, tn[:nb.value].copy()

    def pending(self):
        return self.L.trxsig_rxfe_pending(self.h)


class TxBackEnd:
    def __init__(self, ctx, n_streams, lpf_taps, gain=13500.0, device="cuda:0", max_bursts=64, fused=True):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.L = ctx.L
        _bind(self.L)
        self.S = n_streams
        self.sps = ctx.sps
        self.dev = torch.device(device)
        lpf = np.ascontiguousarray(lpf_taps, np.float32)
        h = C.c_void_p()
        ctx._chk(self.L.trxsig_txbe_create(C.byref(h), ctx.h, n_streams, max_bursts, lpf.ctypes.data, lpf.size, float(gain)),
                 "trxsig_txbe_create")
        self.h = h
        ctx._chk(self.L.trxsig_txbe_set_fused(h, int(fused)), "trxsig_txbe_set_fused")

    def close(self):
        if self.h:
            self.L.trxsig_txbe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push_bursts(self, bits, guard, gain=None):
        """bits: uint8 [S, nb, 148] (numpy or device tensor); guard: int32 [nb] guard symbols per burst (same schedule on
        every stream); gain: optional float32 [S, nb] (addRadioVector's power scaling)."""
        torch = self.torch
        d_bits = bits if torch.is_tensor(bits) else torch.as_tensor(np.ascontiguousarray(bits, np.uint8)).to(self.dev)
        assert d_bits.shape[0] == self.S and d_bits.shape[2] == 148
        guard = np.ascontiguousarray(guard, np.int32)
        d_gain = None
        if gain is not None:
            d_gain = gain if torch.is_tensor(gain) else torch.as_tensor(np.ascontiguousarray(gain, np.float32)).to(self.dev)
        self.ctx._chk(self.L.trxsig_txbe_push_bursts(self.h, d_bits.contiguous().data_ptr(), guard.ctypes.data,
                                                     None if d_gain is None else d_gain.contiguous().data_ptr(), d_bits.shape[1]),
                      "trxsig_txbe_push_bursts")
        self._keep = (d_bits, d_gain)

    def pop_samples(self):
        """int16 tensor [S, n, 2] for the radio (a strided view of the library's output buffer, valid until the next pop),
        or None while less than one chunk is buffered."""
        p = C.c_void_p(); stride = C.c_int64(); n = C.c_int()
        self.ctx._chk(self.L.trxsig_txbe_pop(self.h, C.byref(p), C.byref(stride), C.byref(n)), "trxsig_txbe_pop")
        if n.value == 0:
            return None
        # (the buffer is the back end's own, the same every pop: wrapped once -- torch.as_tensor on a __cuda_array_interface__
        #  object asks the runtime about the pointer, which took ~0.4 ms per call whenever the device was busy)
        key = (p.value, stride.value)
        if getattr(self, "_iq_key", None) != key:
            self._iq_full = self.torch.as_tensor(_DevView(p.value, (self.S, stride.value, 2), "<i2"), device=self.dev)
            self._iq_key = key
        return self._iq_full[:, :n.value]

    def pending(self):
        return self.L.trxsig_txbe_pending(self.h)
