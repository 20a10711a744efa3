"""The sigProcLib.h-compatible C++ facade (include/sigProcLib_trx.h): it compiles against the C-ABI with a
plain host compiler (CPU check) and, on the GPU box, the reference's test call sequence written with the
reference's own function names recovers every bit."""
import os
import subprocess

import pytest

import _pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_smoke(tmp_path, src="facade_smoke.cpp", defines=()):
    m = _pkg.load()
    if not os.path.exists(m.LIB_PATH):
        m.build()
    exe = str(tmp_path / os.path.splitext(src)[0])
    libdir = os.path.dirname(m.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Werror"] + ["-D" + d for d in defines] +
                          ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", src), "-L", libdir, "-ltrxsig",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_facade_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build_smoke(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "no gfx950" in r.stderr.lower() or "no hip device" in r.stderr.lower()


@pytest.mark.gpu
@pytest.mark.parametrize("sps", [1, 4])
def test_facade_loopback(tmp_path, sps):
    exe = build_smoke(tmp_path)
    r = subprocess.run([exe, str(sps)], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "normal burst: 15" in r.stdout and " 0 bit errors" in r.stdout


def test_config1_and_52m_facades_compile(tmp_path):
    """The config-1 driver and the Transceiver52M signatures (TRXFACADE_52M) compile with a plain host compiler."""
    build_smoke(tmp_path, "facade_config1.cpp")
    src = tmp_path / "f52.cpp"
    src.write_text("""#include "sigProcLib_trx.h"
int main() {
  signalVector a(10), b(a, a);                       // the concatenating constructor (radioInterface.cpp:141,244)
  Vector<complex> seg = b.segment(2, 5); seg.fill(complex(1, 2));
  a.segmentCopyTo(b, 1, 3);
  complex amp; float toa = 0;
  analyzeTrafficBurst(a, 0, 3.0f, 1, &amp, &toa, 4u);                         // Transceiver52M/sigProcLib.h:295-305
  SoftVector *s = demodulateBurst(a, a, 1, amp, toa); delete s;               // :324-328 (in place)
  return b.size() == 20 && b[3].r == 1.0f ? 0 : 1;
}
""")
    exe = str(tmp_path / "f52")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-DTRXFACADE_52M", "-I", os.path.join(ROOT, "include"), str(src),
                           "-L", os.path.dirname(_pkg.load().LIB_PATH), "-ltrxsig", "-Wl,-rpath," + os.path.dirname(_pkg.load().LIB_PATH),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    assert subprocess.run([exe]).returncode == 0             # (no library call is reached without sigProcLibSetup)


@pytest.mark.gpu
def test_config1_call_sequence_through_the_facade(tmp_path, golden):
    """BASELINE config 1 on the GPU: Transceiver/sigProcLibTest.cpp's call sequence through include/sigProcLib_trx.h, every
    intermediate value-exact against the vectors captured from the real reference (tests/golden/config1_loopback.npz)."""
    import numpy as np
    g = golden("config1_loopback.npz"); lp = golden("resample.npz")
    exe = build_smoke(tmp_path, "facade_config1.cpp")
    d = tmp_path / "c1"; d.mkdir()
    with open(d / "in.bin", "wb") as f:
        f.write(g["bits"].astype(np.int8).tobytes()); f.write(g["rach_bits"].astype(np.int8).tobytes())
        f.write(lp["rcvLPF_651_raw"].astype(np.float32).tobytes()); f.write(lp["sendLPF_961_raw"].astype(np.float32).tobytes())
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    want_errs = (int(((g["soft"][:148] > 0.5) != g["bits"]).sum()), int(((g["eq_soft"][:148] > 0.5) != g["bits"]).sum()))
    assert "slicer bit errors %d, DFE bit errors %d" % want_errs in r.stdout      # the reference's own counts (1 and 2: the
    #                                       6.932-sample delay of :125 pushes the burst's tail out of the 149 samples)

    def rd(name, cplx=True):
        a = np.fromfile(d / (name + ".bin"), np.float32)
        return a.view(np.complex64) if cplx else a
    assert rd("energy", False)[0] == g["energy"]
    for name in ("mod", "up", "dn", "autocorr", "delayed", "rx", "noise", "rx_noisy", "chan", "dfe_w", "dfe_b"):
        got, want = rd(name), g[name]
        assert got.shape == want.shape and np.array_equal(got, want), name
    for name in ("soft", "eq_soft", "lpf_tx", "lpf_rx"):
        got, want = rd(name, False), g[name]
        assert got.shape == want.shape and np.array_equal(got, want), name
    det = rd("det", False)
    assert det[0] == float(g["ok"]) == 1.0 and complex(det[1], det[2]) == complex(g["amp"]) and det[3] == g["toa"] and det[4] == g["chan_off"]
    rach = rd("rach", False)
    assert rach[0] == float(g["rach_ok"]) and complex(rach[1], rach[2]) == complex(g["rach_amp"]) and rach[3] == g["rach_toa"]
    assert np.array_equal(rd("rach_x"), g["rach_x"])
