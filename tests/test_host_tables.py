"""Host-side (no GPU) checks of the product library: it loads, exports every symbol include/trxsig.h and include/trxsig_transceiver.h
declares, refuses to create a context without a gfx950 device (no CPU fallback), and its init-time
table construction is bit-identical to the reference's tables (tests/golden/tables.npz)."""
import os
import re

import numpy as np
import pytest

import _pkg
from util import assert_beq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    m = _pkg.load()
    if not os.path.exists(m.LIB_PATH):
        m.build()
    return m


def test_exports_every_declared_symbol(pkg):
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("trxsig.h", "trxsig_transceiver.h", "trxsig_frontend.h", "trxsig_trxgroup.h"))
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(trxsig_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 90 and "trxsig_rxfe_push" in names and "trxsig_convolve_batch" in names and "trxsig_trx_pull_radio_vector" in names and "trxsig_fec_tch_decode_batch" in names and "trxsig_trxgroup_pull" in names
    L = pkg.lib()
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert L.trxsig_abi_version() == pkg.ABI_VERSION == 2


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.TrxSigError):
        pkg.TrxSig(4)


@pytest.mark.parametrize("sps", [1, 2, 4])
def test_tables_match_reference(pkg, golden, sps):
    g = golden("tables.npz")
    blob = pkg.build_tables_host(sps)
    dt = pkg.tables_dtype()
    assert dt.itemsize == blob.size == pkg.lib().trxsig_tables_bytes(sps)
    T = blob.view(dt)[0]
    p = "sps%d_" % sps
    assert T["sps"] == sps and T["bytes"] == blob.size
    assert_beq(T["cosT"][:1025], g["cosT"]); assert_beq(T["sinT"][:1025], g["sinT"])
    assert_beq(T["rot"][:157 * sps], g[p + "rot"]); assert_beq(T["rev"][:157 * sps], g[p + "rev"])
    assert_beq(T["pulse"][:2 * sps + 1], g[p + "pulse"])
    assert_beq(np.ascontiguousarray(T["mid"][:, :16 * sps]), g[p + "mid"])
    assert_beq(T["mid_toa"], g[p + "mid_toa"]); assert_beq(T["mid_gain"], g[p + "mid_gain"])
    assert_beq(T["rach"][:41 * sps], g[p + "rach"])
    assert T["rach_toa"] == g[p + "rach_toa"] and T["rach_gain"] == g[p + "rach_gain"]
    # derived tables
    assert_beq(np.ascontiguousarray(T["mid_ctap"]), np.conj(g[p + "mid"][:, ::sps]).astype(np.complex64))
    grid = g["sinc_grid"]                      # sinc(pi*d), d = k/512, k = -5632..5632
    f = np.arange(512)[:, None]; j = np.arange(21)[None, :]
    k = (j - 10) * 512 - f                      # d = (j-10) - f/512
    assert_beq(np.ascontiguousarray(T["sinc_grid"][:, :21]), grid[k + 11 * 512])
    assert not T["sinc_grid"][:, 21:].any()
    # zero taps of the unit-pulse midamble really are zeros (the kernels skip them)
    mask = np.ones(16 * sps, bool); mask[::sps] = False
    assert not g[p + "mid"][:, mask].any()


def test_table_construction_under_sanitizers(tmp_path):
    """trxsig_tablegen.cpp (plain host C++) built with AddressSanitizer + UBSan by g++: tables for sps 1, 2, 4 are
    built and validated without a finding."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "tg_asan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-I", os.path.join(root, "openbts-ttsou_amd", "csrc"), "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "tablegen_sanitize.cpp"),
                           os.path.join(root, "openbts-ttsou_amd", "csrc", "trxsig_tablegen.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.count(" ok checksum ") == 3, (r.stdout, r.stderr)


def test_rach_error_bound_is_small_against_the_sequence():
    """The access-burst detector's error bar (trxsig_tables_rach_error_bound, host computation): for the tables the
    library builds it must stay near 6e-5 of the sequence norm.  A mismatch between the implied sequence of the
    approximate pass and the table sequence would not change any result (everything doubtful is recomputed exactly)
    but would silently send every burst down the slow exact route -- this is where it would show."""
    import ctypes as C
    import _pkg
    pkg = _pkg.load(); L = pkg.lib()
    L.trxsig_tables_rach_error_bound.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    for sps in (1, 2, 4):
        blob = pkg.build_tables_host(sps)
        b, n = C.c_float(), C.c_float()
        assert L.trxsig_tables_rach_error_bound(blob.ctypes.data, blob.size, C.byref(b), C.byref(n)) == 0
        assert 3.2e-5 < b.value / n.value < 8e-5, (sps, b.value, n.value)
        bad = blob.copy(); bad[100] ^= 1
        assert L.trxsig_tables_rach_error_bound(bad.ctypes.data, bad.size, C.byref(b), C.byref(n)) != 0


def test_product_library_carries_the_defaults_only(pkg):
    """The alternates that measured slower (single-launch normal-burst kernels, the other peak kernels, the exact-at-every-lag
    RACH route) are compiled into libtrxsig_tune.so only: the product library neither contains their kernels nor lets
    trxsig_set_tuning select them."""
    prod = open(pkg.LIB_PATH, "rb").read()
    tune = open(pkg.TUNE_LIB_PATH, "rb").read()
    # kernel symbols as the code objects carry them (Itanium names of template instantiations: <len><name>IL...; the plain
    # strings also appear in the profiler's name table, which both libraries share)
    for k in (b"14k_normal_fusedIL", b"13k_normal_quadIL", b"14k_normal_chainIL", b"11k_tsc_peak8IL", b"10k_tsc_peakIL", b"11k_rach_corrIL"):
        assert k not in prod, k
        assert k in tune, k
    for k in (b"10k_tsc_corrIL", b"11k_tsc_peak2IL", b"7k_demodIL", b"12k_rach_frontIL", b"12k_rach_peak2IL", b"11k_rach_fastIL", b"10k_resampleIL"):
        assert k in prod, k
    assert pkg.lib().trxsig_tuning_build() == 0 and pkg.tune_lib().trxsig_tuning_build() == 1
    assert pkg.lib().trxsig_abi_version() == pkg.tune_lib().trxsig_abi_version()


SIGPROCLIB_H_FUNCTIONS = (   # every free function Transceiver/sigProcLib.h:101-384 declares (33 names)
    "dB", "dBinv", "vectorNorm2", "vectorPower", "sigProcLibSetup", "sigProcLibDestroy", "convolve", "generateGSMPulse",
    "frequencyShift", "correlate", "vectorSlicer", "modulateBurst", "sinc", "delayVector", "addVector", "gaussianNoise",
    "interpolatePoint", "peakDetect", "scaleVector", "offsetVector", "generateMidamble", "generateRACHSequence", "energyDetect",
    "detectRACHBurst", "analyzeTrafficBurst", "decimateVector", "demodulateBurst", "createLPF", "polyphaseResampleVector",
    "resampleVector", "designDFE", "equalizeBurst")


def test_facade_covers_the_whole_sigproclib_header(tmp_path):
    """include/sigProcLib_trx.h defines every free function of the reference's sigProcLib.h under its own name, in the
    global namespace (a translation unit that takes the address of each one compiles), and the header's own list of names
    is the one above -- checked against the reference header where it is present (build container)."""
    import subprocess
    assert len(SIGPROCLIB_H_FUNCTIONS) == 32 and len(set(SIGPROCLIB_H_FUNCTIONS)) == 32      # + the `complex` typedef = 33 names
    ref = "/root/reference/Transceiver/sigProcLib.h"
    if os.path.exists(ref):
        text = "".join(open(ref).read().splitlines(True)[100:384])
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text)) - {"complex"}
        assert declared == set(SIGPROCLIB_H_FUNCTIONS), declared ^ set(SIGPROCLIB_H_FUNCTIONS)
    src = tmp_path / "names.cpp"
    uses = "\n".join("  (void)sizeof(&%s);" % n if n not in ("convolve", "analyzeTrafficBurst", "demodulateBurst") else "" for n in SIGPROCLIB_H_FUNCTIONS)
    src.write_text('#include "sigProcLib_trx.h"\nint main() {\n%s\n  signalVector a(4), b(2);\n  b.setSymmetry(ABSSYM);\n'
                   '  (void)sizeof(convolve(&a, &b, NULL, NO_DELAY)); complex z; float t;\n'
                   '  (void)sizeof(analyzeTrafficBurst(a, 0, 3.0f, 1, &z, &t)); (void)sizeof(demodulateBurst(a, b, 1, z, t));\n  return 0;\n}\n' % uses)
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
