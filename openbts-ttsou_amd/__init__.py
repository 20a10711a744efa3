"""openbts-ttsou_amd: MI355X-native burst processing for the OpenBTS software transceiver.

The product is the C-ABI shared library `libtrxsig.so` (include/trxsig.h; sources in csrc/).  This
Python module is plumbing only: a ctypes binding used by tests/ and bench.py to hand torch device
buffers and streams to the library.  There is no CPU fallback anywhere in this package: loading
fails loudly if the library is missing, and creating a context fails without a gfx950 GPU.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRXSIG_LIB", os.path.join(_HERE, "libtrxsig.so"))   # TRXSIG_LIB: tuning builds

F_ENERGY, F_DETECT, F_BADLEN = 1, 2, 128
SOFT_EXACT, SOFT_TOLERANCE = 0, 1                # trxsig_set_soft_mode
ABI_VERSION = 2                                  # TRXSIG_ABI_VERSION of include/trxsig.h


class TrxSigError(RuntimeError):
    pass


class C32(C.Structure):
    """trxsig_c32 by value."""
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


def build(verbose=False):
    """Compile csrc/ into libtrxsig.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-j8", "-C", os.path.join(_HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


TUNE_LIB_PATH = os.path.join(_HERE, "libtrxsig_tune.so")
_lib = None
_libs = {}


def tune_lib():
    """libtrxsig_tune.so: the product's code plus the alternates that measured slower (trxsig_set_tuning) -- for A/B
    measurements and the tests that keep those alternates bit-identical.  Same C-ABI; both libraries can be loaded at once
    (linked -Bsymbolic)."""
    return lib(TUNE_LIB_PATH)


def lib(path=None):
    """The loaded libtrxsig.so.  torch (if used) must be imported first so that the library binds to
    the HIP runtime torch already loaded (same libamdhip64.so.7 soname) and device pointers are
    shared between the two."""
    global _lib
    if path is not None:
        if path not in _libs:
            saved = _lib
            _lib = None
            try:
                _libs[path] = _load(path)
            finally:
                _lib = saved
        return _libs[path]
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def _load(path):
    global _lib
    if True:
        if not os.path.exists(path):
            raise TrxSigError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C openbts-ttsou_amd/csrc` (there is no CPU fallback)" % os.path.basename(path))
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
        L.trxsig_abi_version.restype = i32
        L.trxsig_create.argtypes = [C.POINTER(vp), i32, i32]
        L.trxsig_create_from_tables.argtypes = [C.POINTER(vp), i32, vp, C.c_size_t]
        L.trxsig_destroy.argtypes = [vp]; L.trxsig_destroy.restype = None
        L.trxsig_sps.argtypes = [vp]; L.trxsig_device.argtypes = [vp]
        L.trxsig_set_stream.argtypes = [vp, vp]
        L.trxsig_synchronize.argtypes = [vp]
        L.trxsig_last_error.argtypes = [vp]; L.trxsig_last_error.restype = C.c_char_p
        L.trxsig_reserve.argtypes = [vp, i32]
        L.trxsig_tables_bytes.argtypes = [i32]; L.trxsig_tables_bytes.restype = C.c_size_t
        L.trxsig_tables_device.argtypes = [vp]; L.trxsig_tables_device.restype = vp
        L.trxsig_tables_build_host.argtypes = [i32, vp, C.c_size_t]
        L.trxsig_tables_export.argtypes = [vp, vp, C.c_size_t]
        L.trxsig_detect_demod_normal_batch.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp,
                                                       i32, i32]
        L.trxsig_detect_demod_rach_batch.argtypes = [vp, vp, vp, vp, i32, f32, f32, vp, vp, vp, vp, vp, vp, i32, i32]
        L.trxsig_detect_demod_normal_host.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp, vp, vp, i32, i32]
        L.trxsig_detect_demod_rach_host.argtypes = [vp, vp, vp, vp, i32, f32, f32, vp, vp, vp, vp, vp, i32, i32]
        L.trxsig_demodulate_batch.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32]
        L.trxsig_equalize_normal_batch.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, i32, i32, vp, vp, vp, vp, vp,
                                                   vp, vp, i32, i32]
        L.trxsig_equalize_normal_batch_fmt.argtypes = [vp, vp, i32, vp, vp, i32, i32, f32, f32, i32, i32, vp, vp, vp, vp, vp,
                                                       vp, vp, i32, i32]
        L.trxsig_modulate_batch.argtypes = [vp, vp, vp, vp, i32, vp, vp]
        L.trxsig_modulate_host.argtypes = [vp, vp, vp, vp, i32, vp, vp, C.c_int64]
        L.trxsig_resample_batch.argtypes = [vp, vp, i32, C.c_int64, i32, i32, i32, vp, i32, vp, C.c_int64]
        L.trxsig_resample_out_len.argtypes = [i32, i32, i32]
        L.trxsig_unpack_int16.argtypes = [vp, vp, C.c_int64, i32, vp]
        L.trxsig_pack_int16.argtypes = [vp, vp, C.c_int64, vp]
        L.trxsig_unpack_half.argtypes = [vp, vp, C.c_int64, vp]
        L.trxsig_pack_int16_scaled.argtypes = [vp, vp, C.c_int64, C.c_float, vp]
        L.trxsig_fec_xcch_decode_batch.argtypes = [vp, vp, i32, i32, i32, vp, vp]
        L.trxsig_fec_rach_decode_batch.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
        L.trxsig_channel_estimate_batch.argtypes = [vp, vp, vp, vp, i32, i32, C.c_float, i32, i32, vp, vp, vp, vp, vp]
        L.trxsig_estimate_dfe_batch.argtypes = [vp, vp, vp, vp, i32, i32, C.c_float, C.c_float, C.c_float, i32, i32, vp, vp, vp, vp, vp, vp]
        L.trxsig_design_dfe_batch.argtypes = [vp, vp, vp, vp, i32, vp, vp]
        L.trxsig_fec_xcch_encode_batch.argtypes = [vp, vp, i32, i32, vp]
        L.trxsig_fec_tch_decode_batch.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
        L.trxsig_fec_viterbi_batch.argtypes = [vp, vp, i32, C.c_int64, i32, vp, C.c_int64]
        # sigProcLib.h's free-standing primitives
        L.trxsig_convolve_out_len.argtypes = [i32, i32, i32, i32]
        L.trxsig_convolve_batch.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp]
        L.trxsig_convolve_host.argtypes = [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp, i32]
        L.trxsig_delay_vector_batch.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp]
        L.trxsig_delay_vector_host.argtypes = [vp, vp, i32, f32, i32]
        L.trxsig_interpolate_point_batch.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp]
        L.trxsig_interpolate_point_host.argtypes = [vp, vp, i32, f32, i32, vp]
        L.trxsig_peak_detect_batch.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
        L.trxsig_peak_detect_host.argtypes = [vp, vp, i32, vp, vp, vp]
        L.trxsig_scale_vector_batch.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32]
        L.trxsig_gmsk_rotate_batch.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32]
        L.trxsig_vector_slicer_batch.argtypes = [vp, vp, vp, vp, i32, i32]
        L.trxsig_decimate_batch.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
        L.trxsig_elementwise_host.argtypes = [vp, i32, vp, i32, C32, i32]
        L.trxsig_decimate_host.argtypes = [vp, vp, i32, i32, vp]
        L.trxsig_energy_detect_batch.argtypes = [vp, vp, vp, vp, i32, C.c_uint, i32, f32, vp, vp]
        L.trxsig_energy_detect_host.argtypes = [vp, vp, i32, C.c_uint, i32, f32, vp]
        L.trxsig_db.argtypes = [f32]; L.trxsig_db.restype = f32
        L.trxsig_dbinv.argtypes = [f32]; L.trxsig_dbinv.restype = f32
        L.trxsig_sinc_host.argtypes = [vp, f32, C.POINTER(f32)]
        L.trxsig_gaussian_noise_host.argtypes = [i32, f32, C32, vp]
        L.trxsig_vector_norm2_batch.argtypes = [vp, vp, vp, vp, i32, vp, vp]
        L.trxsig_vector_norm2_host.argtypes = [vp, vp, i32, C.POINTER(f32), C.POINTER(f32)]
        L.trxsig_frequency_shift_batch.argtypes = [vp, vp, vp, vp, i32, vp, vp, i32, vp, vp]
        L.trxsig_frequency_shift_host.argtypes = [vp, vp, i32, f32, f32, i32, vp, C.POINTER(f32)]
        L.trxsig_add_vector_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32]
        L.trxsig_add_vector_host.argtypes = [vp, vp, i32, vp, i32]
        L.trxsig_offset_vector_batch.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32]
        L.trxsig_resample_linear_out_len.argtypes = [i32, f32]
        L.trxsig_resample_linear_batch.argtypes = [vp, vp, vp, vp, i32, f32, vp, vp, vp]
        L.trxsig_resample_linear_host.argtypes = [vp, vp, i32, f32, C32, vp, i32]
        L.trxsig_timer_start.argtypes = [vp]
        L.trxsig_timer_stop.argtypes = [vp, C.POINTER(f32)]
        L.trxsig_kernel_name.argtypes = [i32]; L.trxsig_kernel_name.restype = C.c_char_p
        L.trxsig_profile_enable.argtypes = [vp, i32]
        L.trxsig_set_tuning.argtypes = [vp, i32, i32]
        L.trxsig_set_soft_mode.argtypes = [vp, i32]
        L.trxsig_get_soft_mode.argtypes = [vp]
        L.trxsig_profile_collect.argtypes = [vp, C.POINTER(f32), C.POINTER(i32)]
        L.trxsig_profile_collect_n.argtypes = [vp, i32, C.POINTER(f32), C.POINTER(i32)]
        L.trxsig_kernel_count.restype = i32
        if L.trxsig_abi_version() != ABI_VERSION:
            raise TrxSigError("%s speaks ABI %d, this binding was written for %d" % (path, L.trxsig_abi_version(), ABI_VERSION))
        L.trxsig_tables_validate_host.argtypes = [vp, C.c_size_t]
        return L


def tables_dtype():
    """numpy view of the TrxTables blob (csrc/trxsig_tables.h)."""
    import numpy as np
    return np.dtype([("magic", "<u4"), ("version", "<u4"), ("sps", "<u4"), ("bytes", "<u4"), ("checksum", "<u4"),
                     ("pad0", "<u4", 3), ("cosT", "<f4", 1028), ("sinT", "<f4", 1028), ("rot", "<c8", 628),
                     ("rev", "<c8", 628), ("pulse", "<f4", 12), ("mid", "<c8", (8, 64)), ("mid_toa", "<f4", 8),
                     ("mid_gain", "<c8", 8), ("rach", "<c8", 164), ("rach_toa", "<f4"), ("pad1", "<f4"),
                     ("rach_gain", "<c8"), ("mid_ctap", "<c8", (8, 16)), ("pad2", "<f4", 16),
                     ("sinc_grid", "<f4", (512, 32))])


def build_tables_host(sps):
    """The constant-table blob built on the host (no device needed): uint8 array."""
    import numpy as np
    n = lib().trxsig_tables_bytes(sps)
    buf = np.zeros(n, np.uint8)
    rc = lib().trxsig_tables_build_host(sps, buf.ctypes.data, n)
    if rc != 0:
        raise TrxSigError("trxsig_tables_build_host(%d) failed (%d)" % (sps, rc))
    return buf


def tables_valid(blob):
    """True if a host uint8 array holds a valid table blob (header + checksum)."""
    return lib().trxsig_tables_validate_host(blob.ctypes.data, blob.size) == 0


def _ptr(t):
    """torch tensor / int / None -> device address."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


class TrxSig:
    """One library context = one GPU + one stream (trxsig.h)."""

    def __init__(self, sps=4, device=0, tables_blob=None, tuning=False):
        self.L = tune_lib() if tuning else lib()
        self.h = C.c_void_p()
        if tables_blob is None:
            rc = self.L.trxsig_create(C.byref(self.h), device, sps)
        else:
            rc = self.L.trxsig_create_from_tables(C.byref(self.h), device, _ptr(tables_blob),
                                                  tables_blob.numel() * tables_blob.element_size())
        if rc != 0:
            raise TrxSigError("trxsig_create failed (%d): no gfx950 device or bad arguments; "
                              "this library has no CPU fallback" % rc)
        self.sps = self.L.trxsig_sps(self.h)
        self.device = device

    def close(self):
        if self.h:
            self.L.trxsig_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise TrxSigError("%s failed (%d): %s" % (what, rc, self.L.trxsig_last_error(self.h).decode()))

    def set_stream(self, stream_handle):
        self._chk(self.L.trxsig_set_stream(self.h, stream_handle), "trxsig_set_stream")

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        self._chk(self.L.trxsig_synchronize(self.h), "trxsig_synchronize")

    def reserve(self, max_bursts):
        self._chk(self.L.trxsig_reserve(self.h, max_bursts), "trxsig_reserve")

    def tables_bytes(self):
        return self.L.trxsig_tables_bytes(self.sps)

    def tables_device_ptr(self):
        return self.L.trxsig_tables_device(self.h)

    def tables_export(self):
        import numpy as np
        buf = np.zeros(self.tables_bytes(), np.uint8)
        self._chk(self.L.trxsig_tables_export(self.h, buf.ctypes.data, buf.size), "trxsig_tables_export")
        return buf

    def detect_demod_normal(self, samples, offset, length, tsc, flags, amp, toa, soft, avgpwr=None, hard=None,
                            detect_thresh=3.0, energy_thresh=0.0, nsoft=148, soft_stride=None):
        B = offset.numel() if hasattr(offset, "numel") else len(offset)
        if soft_stride is None:
            soft_stride = soft.shape[-1] if soft is not None and hasattr(soft, "shape") else nsoft
        self._chk(self.L.trxsig_detect_demod_normal_batch(
            self.h, _ptr(samples), _ptr(offset), _ptr(length), B, tsc, detect_thresh, energy_thresh,
            _ptr(flags), _ptr(amp), _ptr(toa), _ptr(avgpwr), _ptr(soft), _ptr(hard), nsoft, soft_stride),
            "trxsig_detect_demod_normal_batch")

    def detect_demod_rach(self, samples, offset, length, flags, amp, toa, soft, avgpwr=None, hard=None,
                          detect_thresh=5.0, energy_thresh=0.0, nsoft=148, soft_stride=None):
        B = offset.numel() if hasattr(offset, "numel") else len(offset)
        if soft_stride is None:
            soft_stride = soft.shape[-1] if soft is not None and hasattr(soft, "shape") else nsoft
        self._chk(self.L.trxsig_detect_demod_rach_batch(
            self.h, _ptr(samples), _ptr(offset), _ptr(length), B, detect_thresh, energy_thresh,
            _ptr(flags), _ptr(amp), _ptr(toa), _ptr(avgpwr), _ptr(soft), _ptr(hard), nsoft, soft_stride),
            "trxsig_detect_demod_rach_batch")

    def detect_demod_host(self, x, off, length, tsc=None, detect_thresh=None, energy_thresh=0.0, nsoft=148):
        """Host-buffer convenience call (numpy in/out, PCIe-inclusive): tsc=None -> RACH path."""
        import numpy as np
        x = np.ascontiguousarray(x, np.complex64); off = np.ascontiguousarray(off, np.int32)
        length = np.ascontiguousarray(length, np.int32)
        B = len(off)
        flags = np.zeros(B, np.uint8); amp = np.zeros(B, np.complex64); toa = np.zeros(B, np.float32)
        pwr = np.zeros(B, np.float32); soft = np.zeros((B, nsoft), np.float32)
        a = lambda v: v.ctypes.data
        if tsc is None:
            rc = self.L.trxsig_detect_demod_rach_host(self.h, a(x), a(off), a(length), B,
                                                      5.0 if detect_thresh is None else detect_thresh, energy_thresh,
                                                      a(flags), a(amp), a(toa), a(pwr), a(soft), nsoft, nsoft)
        else:
            rc = self.L.trxsig_detect_demod_normal_host(self.h, a(x), a(off), a(length), B, tsc,
                                                        3.0 if detect_thresh is None else detect_thresh,
                                                        energy_thresh, a(flags), a(amp), a(toa), a(pwr), a(soft),
                                                        nsoft, nsoft)
        self._chk(rc, "trxsig_detect_demod_*_host")
        return dict(flags=flags, amp=amp, toa=toa, pwr=pwr, soft=soft)

    def demodulate(self, samples, offset, length, amp, toa, soft, enable=None, hard=None, nsoft=148,
                   soft_stride=None):
        B = offset.numel()
        if soft_stride is None:
            soft_stride = soft.shape[-1]
        self._chk(self.L.trxsig_demodulate_batch(self.h, _ptr(samples), _ptr(offset), _ptr(length), B, _ptr(amp),
                                                 _ptr(toa), _ptr(enable), _ptr(soft), _ptr(hard), nsoft,
                                                 soft_stride), "trxsig_demodulate_batch")

    def equalize_normal(self, samples, offset, length, tsc, flags, amp, toa, soft, w=None, b=None, hard=None,
                        detect_thresh=3.0, energy_thresh=0.0, variant52m=True, max_toa=4, nsoft=148,
                        soft_stride=None, fp16=False):
        """fp16=True: `samples` holds half-precision I/Q pairs (read directly by the kernels)."""
        if soft_stride is None:
            soft_stride = soft.shape[-1]
        self._chk(self.L.trxsig_equalize_normal_batch_fmt(
            self.h, _ptr(samples), int(bool(fp16)), _ptr(offset), _ptr(length), offset.numel(), tsc, detect_thresh, energy_thresh,
            int(variant52m), max_toa, _ptr(flags), _ptr(amp), _ptr(toa), _ptr(w), _ptr(b), _ptr(soft), _ptr(hard),
            nsoft, soft_stride), "trxsig_equalize_normal_batch")

    def modulate(self, bits, guard, out, out_offset, gain=None):
        """bits [B,148] uint8, guard [B] int32, out packed complex (as float32 pairs), out_offset [B] int32."""
        self._chk(self.L.trxsig_modulate_batch(self.h, _ptr(bits), _ptr(guard), _ptr(gain), guard.numel(), _ptr(out),
                                               _ptr(out_offset)), "trxsig_modulate_batch")

    def modulate_host(self, bits, guard, gain=None):
        import numpy as np
        bits = np.ascontiguousarray(bits, np.uint8); guard = np.ascontiguousarray(guard, np.int32)
        B = len(guard)
        length = (self.sps * (148 + guard)).astype(np.int32)
        off = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.int32)
        out = np.zeros(int(length.sum()), np.complex64)
        g = None if gain is None else np.ascontiguousarray(gain, np.float32)
        self._chk(self.L.trxsig_modulate_host(self.h, bits.ctypes.data, guard.ctypes.data,
                                              None if g is None else g.ctypes.data, B, out.ctypes.data,
                                              off.ctypes.data, out.size), "trxsig_modulate_host")
        return out, off, length

    # ---- sigProcLib.h's free-standing primitives, single-vector host forms (numpy in / out) ----
    # ---- the rest of sigProcLib.h (host forms) ----
    def db(self, x): return self.L.trxsig_db(float(x))
    def dbinv(self, x): return self.L.trxsig_dbinv(float(x))

    def sinc_host(self, x):
        v = C.c_float()
        self._chk(self.L.trxsig_sinc_host(self.h, float(x), C.byref(v)), "trxsig_sinc_host")
        return v.value

    def gaussian_noise_host(self, seed, length, variance=1.0, mean=0j):
        import numpy as np
        C.CDLL(None).srand(int(seed))
        out = np.zeros(length, np.complex64)
        self._chk(self.L.trxsig_gaussian_noise_host(length, float(variance), C32(float(np.real(mean)), float(np.imag(mean))), out.ctypes.data),
                  "trxsig_gaussian_noise_host")
        return out

    def vector_norm2_host(self, x):
        import numpy as np
        x = np.ascontiguousarray(x, np.complex64); e = C.c_float(); p = C.c_float()
        self._chk(self.L.trxsig_vector_norm2_host(self.h, x.ctypes.data, x.size, C.byref(e), C.byref(p)), "trxsig_vector_norm2_host")
        return np.float32(e.value), np.float32(p.value)

    def frequency_shift_host(self, x, freq, start_phase=0.0, real_only=False):
        import numpy as np
        x = np.ascontiguousarray(x, np.complex64); y = np.zeros_like(x); fin = C.c_float()
        self._chk(self.L.trxsig_frequency_shift_host(self.h, x.ctypes.data, x.size, float(freq), float(start_phase), int(real_only),
                                                     y.ctypes.data, C.byref(fin)), "trxsig_frequency_shift_host")
        return y, np.float32(fin.value)

    def add_vector_host(self, x, y):
        import numpy as np
        x = np.array(x, np.complex64, copy=True); y = np.ascontiguousarray(y, np.complex64)
        self._chk(self.L.trxsig_add_vector_host(self.h, x.ctypes.data, x.size, y.ctypes.data, y.size), "trxsig_add_vector_host")
        return x

    def resample_linear_host(self, x, exp_factor, end_point=0j):
        import numpy as np
        x = np.ascontiguousarray(x, np.complex64)
        n = self.L.trxsig_resample_linear_out_len(x.size, float(exp_factor))
        if n < 0:
            return None
        out = np.zeros(max(n, 1), np.complex64)
        rc = self.L.trxsig_resample_linear_host(self.h, x.ctypes.data, x.size, float(exp_factor),
                                                C32(float(np.real(end_point)), float(np.imag(end_point))), out.ctypes.data, out.size)
        if rc < 0:
            self._chk(rc, "trxsig_resample_linear_host")
        return out[:rc]

    def convolve_host(self, a, b, span, a_real=False, b_real=False, correlate=False, cust_start=0, cust_len=0, abssym=False):
        import numpy as np
        a = np.ascontiguousarray(a, np.complex64); b = np.ascontiguousarray(b, np.complex64)
        n = self.L.trxsig_convolve_out_len(a.size, b.size, span, cust_len)
        out = np.zeros(max(n, 1), np.complex64)
        rc = self.L.trxsig_convolve_host(self.h, a.ctypes.data, a.size, b.ctypes.data, b.size, span,
                                         int(a_real) | (int(b_real) << 1) | (int(abssym) << 2), int(correlate), cust_start, cust_len, out.ctypes.data, out.size)
        if rc < 0:
            self._chk(rc, "trxsig_convolve_host")
        return out[:rc]

    def delay_vector_host(self, x, delay, real_only=False):
        import numpy as np
        y = np.array(x, np.complex64, copy=True)
        self._chk(self.L.trxsig_delay_vector_host(self.h, y.ctypes.data, y.size, float(delay), int(real_only)), "trxsig_delay_vector_host")
        return y

    def interpolate_point_host(self, x, ix, real_only=False):
        import numpy as np
        x = np.ascontiguousarray(x, np.complex64); out = np.zeros(1, np.complex64)
        self._chk(self.L.trxsig_interpolate_point_host(self.h, x.ctypes.data, x.size, float(ix), int(real_only), out.ctypes.data),
                  "trxsig_interpolate_point_host")
        return out[0]

    def peak_detect_host(self, x):
        import numpy as np
        x = np.ascontiguousarray(x, np.complex64); pk = np.zeros(1, np.complex64); ix = np.zeros(1, np.float32); av = np.zeros(1, np.float32)
        self._chk(self.L.trxsig_peak_detect_host(self.h, x.ctypes.data, x.size, pk.ctypes.data, ix.ctypes.data, av.ctypes.data),
                  "trxsig_peak_detect_host")
        return pk[0], ix[0], av[0]

    def elementwise_host(self, op, x, scale=1.0, real_only=False):
        """op: 0 scaleVector, 1 GMSKRotate, 2 GMSKReverseRotate, 3 vectorSlicer."""
        import numpy as np
        y = np.array(x, np.complex64, copy=True)
        sc = C32(float(np.real(scale)), float(np.imag(scale)))
        self._chk(self.L.trxsig_elementwise_host(self.h, op, y.ctypes.data, y.size, sc, int(real_only)), "trxsig_elementwise_host")
        return y

    def energy_detect_host(self, x, window, thresh, step=1):
        import numpy as np
        x = np.ascontiguousarray(x, np.complex64); av = np.zeros(1, np.float32)
        rc = self.L.trxsig_energy_detect_host(self.h, x.ctypes.data, x.size, int(window), step, float(thresh), av.ctypes.data)
        if rc < 0:
            self._chk(rc, "trxsig_energy_detect_host")
        return bool(rc), av[0]

    def decimate_host(self, x, factor):
        import numpy as np
        x = np.ascontiguousarray(x, np.complex64); out = np.zeros(max(x.size // factor, 1), np.complex64)
        rc = self.L.trxsig_decimate_host(self.h, x.ctypes.data, x.size, factor, out.ctypes.data)
        if rc < 0:
            self._chk(rc, "trxsig_decimate_host")
        return out[:rc]

    def resample_out_len(self, n_in, P, Q):
        return self.L.trxsig_resample_out_len(n_in, P, Q)

    def resample(self, x, n_in, in_stride, S, P, Q, lpf, out, out_stride):
        self._chk(self.L.trxsig_resample_batch(self.h, _ptr(x), n_in, in_stride, S, P, Q, _ptr(lpf), lpf.numel(),
                                               _ptr(out), out_stride), "trxsig_resample_batch")

    def unpack_int16(self, iq, n, out, swap_iq=True):
        self._chk(self.L.trxsig_unpack_int16(self.h, _ptr(iq), n, int(swap_iq), _ptr(out)), "trxsig_unpack_int16")

    def fec_xcch_decode(self, soft, n_blocks, frames, ok, wire=True, soft_stride=None):
        self._chk(self.L.trxsig_fec_xcch_decode_batch(self.h, _ptr(soft), soft_stride or soft.shape[-1], n_blocks,
                                                      int(wire), _ptr(frames), _ptr(ok)), "trxsig_fec_xcch_decode_batch")

    def fec_rach_decode(self, soft, n_bursts, tail_ok, bsic, ra, wire=True, soft_stride=None):
        self._chk(self.L.trxsig_fec_rach_decode_batch(self.h, _ptr(soft), soft_stride or soft.shape[-1], n_bursts,
                                                      int(wire), _ptr(tail_ok), _ptr(bsic), _ptr(ra)),
                  "trxsig_fec_rach_decode_batch")

    def channel_estimate(self, samples, offset, length, tsc, flags, amp, toa, chan_off, chan, detect_thresh=3.0, variant52m=False,
                         max_toa=4):
        self._chk(self.L.trxsig_channel_estimate_batch(self.h, _ptr(samples), _ptr(offset), _ptr(length), offset.numel(), tsc,
                                                       detect_thresh, int(variant52m), max_toa, _ptr(flags), _ptr(amp), _ptr(toa),
                                                       _ptr(chan_off), _ptr(chan)), "trxsig_channel_estimate_batch")

    def estimate_dfe(self, samples, offset, length, tsc, flags, amp, toa, chan_off, w, b, detect_thresh=3.0, snr_thresh=-1.0,
                     snr_value=0.0, variant52m=False, max_toa=4):
        """analyzeTrafficBurst(requestChannel) + scaleVector(chan, 1/amp) + designDFE(., SNR, 7) (Transceiver.cpp:326-347), no energy gate."""
        self._chk(self.L.trxsig_estimate_dfe_batch(self.h, _ptr(samples), _ptr(offset), _ptr(length), offset.numel(), tsc,
                                                   detect_thresh, snr_thresh, snr_value, int(variant52m), max_toa, _ptr(flags), _ptr(amp),
                                                   _ptr(toa), _ptr(chan_off), _ptr(w), _ptr(b)), "trxsig_estimate_dfe_batch")

    def design_dfe(self, chan, snr, w, b, amp=None):
        self._chk(self.L.trxsig_design_dfe_batch(self.h, _ptr(chan), _ptr(amp), _ptr(snr), snr.numel(), _ptr(w), _ptr(b)),
                  "trxsig_design_dfe_batch")

    def fec_xcch_encode(self, frames, n_blocks, tsc, bits):
        self._chk(self.L.trxsig_fec_xcch_encode_batch(self.h, _ptr(frames), n_blocks, tsc, _ptr(bits)), "trxsig_fec_xcch_encode_batch")

    def fec_tch_decode(self, soft, n_bursts, tch, tch_good, stolen, facch=None, facch_ok=None, wire=True, soft_stride=None):
        self._chk(self.L.trxsig_fec_tch_decode_batch(self.h, _ptr(soft), soft_stride or soft.shape[-1], n_bursts, int(wire),
                                                     _ptr(tch), _ptr(tch_good), _ptr(facch), _ptr(facch_ok), _ptr(stolen)),
                  "trxsig_fec_tch_decode_batch")

    def fec_viterbi(self, soft, n_soft, n_blocks, bits, in_stride=None, out_stride=None):
        self._chk(self.L.trxsig_fec_viterbi_batch(self.h, _ptr(soft), n_soft, in_stride or soft.shape[-1], n_blocks,
                                                  _ptr(bits), out_stride or bits.shape[-1]), "trxsig_fec_viterbi_batch")

    def unpack_half(self, iq, n, out):
        self._chk(self.L.trxsig_unpack_half(self.h, _ptr(iq), n, _ptr(out)), "trxsig_unpack_half")

    def pack_int16_scaled(self, x, n, gain, iq):
        self._chk(self.L.trxsig_pack_int16_scaled(self.h, _ptr(x), n, float(gain), _ptr(iq)), "trxsig_pack_int16_scaled")

    def pack_int16(self, x, n, iq):
        self._chk(self.L.trxsig_pack_int16(self.h, _ptr(x), n, _ptr(iq)), "trxsig_pack_int16")

    def set_tuning(self, normal_path=None, rach_path=None, generic_taps=None, spec_peak=None, chain_lag=None,
                   chain_spin=None, demod_beside=None, beside_det_cus=None, cu_layout=None, beside_priority=None,
                   eq_tail=None, eq_dense=None, rxres_wpb=None, rxres_rows=None, chan_tpw=None, group_replay=None):
        """A/B implementation choice (results are bit-identical): see trxsig_set_tuning.  eq_tail / eq_dense / rxres_* / chan_tpw /
        group_replay are LIBRARY-WIDE (every context of the process): restore the default (1 / 4096 / 0 / 1 / 0 / 0) when done."""
        for key, v in ((12, eq_tail), (13, eq_dense), (14, rxres_wpb), (15, rxres_rows), (16, chan_tpw), (17, group_replay)):
            if v is not None:
                self._chk(self.L.trxsig_set_tuning(self.h, key, int(v)), "trxsig_set_tuning")
        if demod_beside is not None:
            self._chk(self.L.trxsig_set_tuning(self.h, 7, int(demod_beside)), "trxsig_set_tuning")
        if beside_det_cus is not None:
            self._chk(self.L.trxsig_set_tuning(self.h, 8, int(beside_det_cus)), "trxsig_set_tuning")
        if beside_priority is not None:
            self._chk(self.L.trxsig_set_tuning(self.h, 11, int(beside_priority)), "trxsig_set_tuning")
        if cu_layout is not None:
            self._chk(self.L.trxsig_set_tuning(self.h, 9, int(cu_layout)), "trxsig_set_tuning")
        if chain_lag is not None:
            self._chk(self.L.trxsig_set_tuning(self.h, 4, int(chain_lag)), "trxsig_set_tuning")
        if chain_spin is not None:
            self._chk(self.L.trxsig_set_tuning(self.h, 5, int(chain_spin)), "trxsig_set_tuning")
        if spec_peak is not None:
            self._chk(self.L.trxsig_set_tuning(self.h, 3, int(spec_peak)), "trxsig_set_tuning")
        if generic_taps is not None:
            self._chk(self.L.trxsig_set_tuning(self.h, 2, int(generic_taps)), "trxsig_set_tuning")
        if normal_path is not None:
            self._chk(self.L.trxsig_set_tuning(self.h, 0, int(normal_path)), "trxsig_set_tuning")
        if rach_path is not None:
            self._chk(self.L.trxsig_set_tuning(self.h, 1, int(rach_path)), "trxsig_set_tuning")

    def set_soft_mode(self, mode):
        """SOFT_EXACT (default: soft bits IEEE-equal to the reference's) or SOFT_TOLERANCE (hard bits, flags, amp, TOA exact;
        soft bits within 7.4e-5 of the reference's -- trxsig_set_soft_mode)."""
        self._chk(self.L.trxsig_set_soft_mode(self.h, int(mode)), "trxsig_set_soft_mode")

    def soft_mode(self):
        return int(self.L.trxsig_get_soft_mode(self.h))

    def profile_enable(self, on=True):
        self._chk(self.L.trxsig_profile_enable(self.h, int(on)), "trxsig_profile_enable")

    def profile_collect(self):
        """{kernel name: (total_ms, launches)} since the last collect (synchronises)."""
        n = self.L.trxsig_kernel_count()
        ms = (C.c_float * n)(); cnt = (C.c_int * n)()
        if self.L.trxsig_profile_collect_n(self.h, n, ms, cnt) < 0:
            self._chk(-1, "trxsig_profile_collect_n")
        return {self.L.trxsig_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(n) if cnt[i]}

    def timer_start(self):
        self._chk(self.L.trxsig_timer_start(self.h), "trxsig_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.L.trxsig_timer_stop(self.h, C.byref(ms)), "trxsig_timer_stop")
        return ms.value


class TrxHost:
    """ctypes view of include/trxsig_transceiver.h: the per-ARFCN Transceiver orchestration (pullRadioVector,
    addRadioVector / pushRadioVector, control commands, UDP datagram codecs) on top of the GPU library."""

    def __init__(self, sps, device=0, start=(0, 0), tsc_leg=0):
        import numpy as np
        self.np = np
        self.L = L = lib()
        vp, i32 = C.c_void_p, C.c_int
        L.trxsig_trx_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32]
        L.trxsig_trx_destroy.argtypes = [vp]; L.trxsig_trx_destroy.restype = None
        L.trxsig_trx_last_error.argtypes = [vp]; L.trxsig_trx_last_error.restype = C.c_char_p
        L.trxsig_trx_control.argtypes = [vp, C.c_char_p, C.c_char_p, i32]
        L.trxsig_trx_expected_corr_type.argtypes = [vp, i32, i32]
        L.trxsig_trx_pull_radio_vector.argtypes = [vp, vp, i32, i32, i32, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.trxsig_trx_encode_rx_datagram.argtypes = [i32, i32, i32, i32, vp, i32, vp]
        L.trxsig_trx_decode_tx_datagram.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), vp]
        L.trxsig_trx_add_radio_vector.argtypes = [vp, vp, i32, i32, i32]
        L.trxsig_trx_push_radio_vector.argtypes = [vp, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]
        L.trxsig_trx_energy_threshold.argtypes = [vp]; L.trxsig_trx_energy_threshold.restype = C.c_double
        L.trxsig_trx_filler_modulus.argtypes = [vp, i32]
        L.trxsig_trx_queue_size.argtypes = [vp]
        L.trxsig_create_lpf_host.argtypes = [vp, i32, C.c_float, vp]
        L.trxsig_trx_set_tsc_leg.argtypes = [vp, i32]
        self.sps = sps
        self.h = vp()
        rc = L.trxsig_trx_create(C.byref(self.h), device, sps, start[0], start[1])
        if rc != 0:
            raise RuntimeError("trxsig_trx_create failed: %d" % rc)
        if tsc_leg:
            self._chk(L.trxsig_trx_set_tsc_leg(self.h, tsc_leg), "trxsig_trx_set_tsc_leg")

    def close(self):
        if self.h:
            self.L.trxsig_trx_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            raise TrxSigError("%s: %d (%s)" % (what, rc, self.L.trxsig_trx_last_error(self.h).decode()))
        return rc

    def control(self, msg):
        buf = C.create_string_buffer(128)
        self._chk(self.L.trxsig_trx_control(self.h, msg.encode(), buf, 128), "trxsig_trx_control")
        return buf.value.decode()

    def expected_corr_type(self, tn, fn):
        return self.L.trxsig_trx_expected_corr_type(self.h, tn, fn)

    def pull_radio_vector(self, x, tn, fn):
        np = self.np
        x = np.ascontiguousarray(x, np.complex64)
        soft = np.zeros(160, np.float32)
        ns, rssi, toa = C.c_int(), C.c_int(), C.c_int()
        rc = self._chk(self.L.trxsig_trx_pull_radio_vector(self.h, x.ctypes.data, len(x), tn, fn, soft.ctypes.data,
                                                           C.byref(ns), C.byref(rssi), C.byref(toa)), "trxsig_trx_pull_radio_vector")
        if rc == 0:
            return None
        return soft[:ns.value].copy(), rssi.value, toa.value

    def encode_rx_datagram(self, tn, fn, rssi, toa, soft):
        np = self.np
        soft = np.ascontiguousarray(soft, np.float32)
        out = np.zeros(158, np.uint8)
        self._chk(self.L.trxsig_trx_encode_rx_datagram(tn, fn, rssi, toa, soft.ctypes.data, len(soft), out.ctypes.data), "encode")
        return out.tobytes()

    def decode_tx_datagram(self, b):
        np = self.np
        a = np.frombuffer(b, np.uint8).copy()
        tn, fn, rssi = C.c_int(), C.c_int(), C.c_int()
        bits = np.zeros(148, np.uint8)
        rc = self.L.trxsig_trx_decode_tx_datagram(a.ctypes.data, len(a), C.byref(tn), C.byref(fn), C.byref(rssi), bits.ctypes.data)
        if rc != 0:
            return None
        return tn.value, fn.value, rssi.value, bits

    def add_radio_vector(self, bits, rssi, tn, fn):
        np = self.np
        bits = np.ascontiguousarray(bits, np.uint8)
        self._chk(self.L.trxsig_trx_add_radio_vector(self.h, bits.ctypes.data, rssi, tn, fn), "trxsig_trx_add_radio_vector")

    def push_radio_vector(self, tn, fn):
        np = self.np
        out = np.zeros(157 * self.sps, np.complex64)
        n, fq = C.c_int(), C.c_int()
        self._chk(self.L.trxsig_trx_push_radio_vector(self.h, tn, fn, out.ctypes.data, C.byref(n), C.byref(fq)), "push")
        return out[:n.value].copy(), bool(fq.value)

    @property
    def energy_threshold(self):
        return self.L.trxsig_trx_energy_threshold(self.h)

    def filler_modulus(self, tn):
        return self.L.trxsig_trx_filler_modulus(self.h, tn)

    def queue_size(self):
        return self.L.trxsig_trx_queue_size(self.h)

    def create_lpf(self, raw, gain):
        np = self.np
        raw = np.ascontiguousarray(raw, np.float32)
        out = np.zeros(len(raw), np.float32)
        self._chk(self.L.trxsig_create_lpf_host(raw.ctypes.data, len(raw), float(gain), out.ctypes.data), "create_lpf")
        return out


TSCLEG_EQUALIZE, TSCLEG_DEMOD = 0, 1


class TrxGroupResult(C.Structure):
    """trxsig_trxgroup_result"""
    _fields_ = [("n_slots", C.c_int), ("n_arfcn", C.c_int), ("n_rows", C.c_int), ("d_row", C.c_void_p), ("d_valid", C.c_void_p),
                ("d_flags", C.c_void_p), ("d_amp", C.c_void_p), ("d_toa", C.c_void_p), ("d_avgpwr", C.c_void_p),
                ("d_threshold", C.c_void_p), ("d_soft", C.c_void_p), ("soft_stride", C.c_int)]


class TrxGroup:
    """ctypes view of include/trxsig_trxgroup.h: S Transceivers' pullRadioVector per call, state machine on the device."""

    def __init__(self, ctx, n_arfcn, tsc_leg=TSCLEG_EQUALIZE, start=(0, 0)):
        import numpy as np
        self.np = np
        self.ctx = ctx
        self.L = L = ctx.L
        vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
        L.trxsig_trxgroup_create.argtypes = [C.POINTER(vp), vp, i32, i32, i32, i32]
        L.trxsig_trxgroup_destroy.argtypes = [vp]; L.trxsig_trxgroup_destroy.restype = None
        L.trxsig_trxgroup_control.argtypes = [vp, i32, C.c_char_p, C.c_char_p, i32]
        L.trxsig_trxgroup_expected_corr_type.argtypes = [vp, i32, i32, i32]
        L.trxsig_trxgroup_pull.argtypes = [vp, vp, i64, i64, i32, i32, i32, i32, C.POINTER(TrxGroupResult)]
        L.trxsig_trxgroup_pull_host.argtypes = [vp, vp, i64, i64, i32, i32, i32, i32]
        L.trxsig_trxgroup_pull_rxfe.argtypes = [vp, vp, vp, i32, i32, C.POINTER(i32), C.POINTER(TrxGroupResult)]
        L.trxsig_trxgroup_collect.argtypes = [vp, vp, vp, vp, vp, vp]
        L.trxsig_trxgroup_energy_threshold.argtypes = [vp, i32, C.POINTER(C.c_double)]
        L.trxsig_trxgroup_set_pipelined.argtypes = [vp, i32]
        L.trxsig_trxgroup_set_beside_rows.argtypes = [vp, i32]
        L.trxsig_trxgroup_set_split_rows.argtypes = [vp, i32]
        L.trxsig_trxgroup_sync.argtypes = [vp]
        L.trxsig_trxgroup_add_bursts.argtypes = [vp, vp, vp, i32]
        L.trxsig_trxgroup_tx_staging.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp)]
        L.trxsig_trxgroup_add_staged.argtypes = [vp, i32]
        L.trxsig_trxgroup_push.argtypes = [vp, i32, i32, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        L.trxsig_trxgroup_push_txbe.argtypes = [vp, vp, i32, i32, i32]
        L.trxsig_trxgroup_tx_queue_size.argtypes = [vp, i32, C.POINTER(i32)]
        self.S = n_arfcn
        self.h = vp()
        rc = L.trxsig_trxgroup_create(C.byref(self.h), ctx.h, n_arfcn, tsc_leg, start[0], start[1])
        if rc != 0:
            raise TrxSigError("trxsig_trxgroup_create failed (%d): %s" % (rc, L.trxsig_last_error(ctx.h).decode()))
        self.n_slots = 0

    def close(self):
        if self.h:
            self.L.trxsig_trxgroup_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            raise TrxSigError("%s: %d (%s)" % (what, rc, self.L.trxsig_last_error(self.ctx.h).decode()))
        return rc

    def control(self, arfcn, msg):
        buf = C.create_string_buffer(128)
        self._chk(self.L.trxsig_trxgroup_control(self.h, arfcn, msg.encode(), buf, 128), "trxsig_trxgroup_control")
        return buf.value.decode()

    def expected_corr_type(self, arfcn, tn, fn):
        return self.L.trxsig_trxgroup_expected_corr_type(self.h, arfcn, tn, fn)

    def pull(self, samples, slot_stride, arfcn_stride, fn, tn, n_slots, burst_len=0):
        """samples: torch complex64-as-float32 device tensor (or a device address)."""
        res = TrxGroupResult()
        self._chk(self.L.trxsig_trxgroup_pull(self.h, _ptr(samples), slot_stride, arfcn_stride, burst_len, fn, tn, n_slots, C.byref(res)),
                  "trxsig_trxgroup_pull")
        self.n_slots = n_slots
        return res

    def pull_rxfe(self, fe, iq, fn):
        """fe: frontend.RxFrontEnd on the same context; iq: int16 device tensor [S, K*864, 2].  Returns (slots completed, result)."""
        iq = iq.contiguous()
        res = TrxGroupResult()
        n = C.c_int()
        self._chk(self.L.trxsig_trxgroup_pull_rxfe(self.h, fe.h, iq.data_ptr(), iq.shape[1] // 864, fn, C.byref(n), C.byref(res)),
                  "trxsig_trxgroup_pull_rxfe")
        fe._keep = iq
        self.n_slots = n.value
        return n.value, res

    # ---- transmit half ----
    def add_bursts(self, datagrams, arfcn):
        """datagrams: uint8 [n, 154] (the 154-byte transmit datagrams, host); arfcn: int32 [n] the ARFCN each arrived for."""
        np = self.np
        d = np.ascontiguousarray(datagrams, np.uint8); a = np.ascontiguousarray(arfcn, np.int32)
        assert d.ndim == 2 and d.shape[1] == 154 and a.shape == (d.shape[0],)
        self._chk(self.L.trxsig_trxgroup_add_bursts(self.h, d.ctypes.data, a.ctypes.data, d.shape[0]), "trxsig_trxgroup_add_bursts")

    def tx_staging(self, n_max):
        """The pinned block to RECEIVE the next batch into: (datagrams uint8 [n_max, 154], arfcn int32 [n_max]) as numpy views of the
        library's memory; valid until add_staged."""
        np = self.np
        pd, pa = C.c_void_p(), C.c_void_p()
        self._chk(self.L.trxsig_trxgroup_tx_staging(self.h, int(n_max), C.byref(pd), C.byref(pa)), "trxsig_trxgroup_tx_staging")
        d = np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_uint8)), shape=(n_max, 154))
        a = np.ctypeslib.as_array(C.cast(pa, C.POINTER(C.c_int32)), shape=(n_max,))
        return d, a

    def add_staged(self, n):
        """The first n datagrams of the staging block: header check on the host, one upload, one kernel."""
        self._chk(self.L.trxsig_trxgroup_add_staged(self.h, int(n)), "trxsig_trxgroup_add_staged")

    def push(self, fn, tn, n_slots, device="cuda:0"):
        """pushRadioVector for n_slots timeslots from (fn, tn): (bits uint8 [S, n, 148], gain float32 [S, n], from_queue uint8 [S, n])
        as torch views of the group's device buffers (valid until the next push)."""
        import torch
        from .frontend import _DevView
        pb, pg, pq = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.L.trxsig_trxgroup_push(self.h, fn, tn, n_slots, C.byref(pb), C.byref(pg), C.byref(pq)), "trxsig_trxgroup_push")
        dev = torch.device(device)
        return (torch.as_tensor(_DevView(pb.value, (self.S, n_slots, 148), "|u1"), device=dev),
                torch.as_tensor(_DevView(pg.value, (self.S, n_slots), "<f4"), device=dev),
                torch.as_tensor(_DevView(pq.value, (self.S, n_slots), "|u1"), device=dev))

    def push_txbe(self, be, fn, tn, n_slots):
        self._chk(self.L.trxsig_trxgroup_push_txbe(self.h, be.h, fn, tn, n_slots), "trxsig_trxgroup_push_txbe")

    def tx_queue_size(self, arfcn):
        dropped = C.c_int()
        n = self._chk(self.L.trxsig_trxgroup_tx_queue_size(self.h, arfcn, C.byref(dropped)), "trxsig_trxgroup_tx_queue_size")
        return n, bool(dropped.value)

    def pull_host(self, x, slot_stride, arfcn_stride, fn, tn, n_slots, burst_len=0):
        np = self.np
        x = np.ascontiguousarray(x, np.complex64)
        self._chk(self.L.trxsig_trxgroup_pull_host(self.h, x.ctypes.data, slot_stride, arfcn_stride, burst_len, fn, tn, n_slots),
                  "trxsig_trxgroup_pull_host")
        self.n_slots = n_slots

    def collect(self, soft=True):
        """dict of host arrays indexed [slot][arfcn]: valid, soft (x148), rssi, timing, threshold."""
        np = self.np
        n = self.n_slots * self.S
        valid = np.zeros(n, np.uint8); rssi = np.zeros(n, np.int32); timing = np.zeros(n, np.int32); thr = np.zeros(n, np.float64)
        sb = np.zeros((n, 148), np.float32) if soft else None
        self._chk(self.L.trxsig_trxgroup_collect(self.h, valid.ctypes.data, sb.ctypes.data if soft else None, rssi.ctypes.data,
                                                 timing.ctypes.data, thr.ctypes.data), "trxsig_trxgroup_collect")
        sh = (self.n_slots, self.S)
        return dict(valid=valid.reshape(sh).astype(bool), soft=None if sb is None else sb.reshape(sh + (148,)), rssi=rssi.reshape(sh),
                    timing=timing.reshape(sh), threshold=thr.reshape(sh))

    def set_pipelined(self, on=True):
        """Large pulls return without joining the side stream the state machine replays on (see trxsig_trxgroup.h)."""
        self._chk(self.L.trxsig_trxgroup_set_pipelined(self.h, 1 if on else 0), "trxsig_trxgroup_set_pipelined")

    def set_beside_rows(self, rows):
        """Pulls with at least `rows` rows replay the state machine on the group's side stream (0 = never, the default)."""
        self._chk(self.L.trxsig_trxgroup_set_beside_rows(self.h, int(rows)), "trxsig_trxgroup_set_beside_rows")

    def set_split_rows(self, rows):
        """Fused pulls with at least `rows` rows and both kinds of burst detect the access bursts beside the normal ones (0 = never)."""
        self._chk(self.L.trxsig_trxgroup_set_split_rows(self.h, int(rows)), "trxsig_trxgroup_set_split_rows")

    def sync(self):
        self._chk(self.L.trxsig_trxgroup_sync(self.h), "trxsig_trxgroup_sync")

    def energy_threshold(self, arfcn):
        v = C.c_double()
        self._chk(self.L.trxsig_trxgroup_energy_threshold(self.h, arfcn, C.byref(v)), "trxsig_trxgroup_energy_threshold")
        return v.value
