"""The sigProcLib.h-compatible C++ facade (include/sigProcLib_trx.h): it compiles against the C-ABI with a
plain host compiler (CPU check) and, on the GPU box, the reference's test call sequence written with the
reference's own function names recovers every bit."""
import os
import subprocess

import pytest

import _pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_smoke(tmp_path):
    m = _pkg.load()
    if not os.path.exists(m.LIB_PATH):
        m.build()
    exe = str(tmp_path / "facade_smoke")
    libdir = os.path.dirname(m.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "facade_smoke.cpp"), "-L", libdir, "-ltrxsig",
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
