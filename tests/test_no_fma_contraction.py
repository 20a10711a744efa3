"""Numerical-contract audit on the generated gfx950 ISA (no GPU needed: hipcc cross-compiles): outside
hipcc's correctly-rounded division / sqrt expansions no kernel may contain a fused multiply-add, except the two
marked kinds (and, in the shared-filter channeliser alone -- an approximate form by construction, off by default, graded with a
tolerance -- the "approx-form" kind; and, in the kernels instantiated for trxsig_set_soft_mode(TRXSIG_SOFT_TOLERANCE) alone, the "soft-tolerance" kind) that tools/asm_stats.py counts apart: the exact-product FMAs of the midamble correlators (a tap
component of exactly +-1: single rounding == separate mul and add) and the FMAs of a steering pass (fma_steer:
approximate correlations that only decide which lags are recomputed with the reference's exact arithmetic), which
may appear in the kernels listed below and nowhere else."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_kernels_have_no_contracted_fma():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "openbts-ttsou_amd", "csrc"), "asm"],
                          stderr=subprocess.DEVNULL)
    out = ""
    for f in ("trxsig_normal", "trxsig_fused", "trxsig_rach", "trxsig_eq", "trxsig_tx", "trxsig_fec", "trxsig_prim", "trxsig_group", "trxsig_chan"):
        f += ".gfx950.s"
        out += subprocess.check_output(["python3", os.path.join(ROOT, "tools", "asm_stats.py"),
                                        os.path.join(ROOT, "openbts-ttsou_amd", "csrc", f)], text=True)
    rows = [l for l in out.splitlines() if "outside a division" in l]
    assert len(rows) >= 23 and any("k_fec_viterbi" in l for l in rows)
    steering_ok = ("k_rach_fast", "k_rach_front")
    for l in rows:
        n = int(re.search(r"outside a division: (\d+)", l).group(1))
        assert n == 0, l
        if "steering fma" in l:
            assert any(k in l for k in steering_ok), l
        if "approx-form fma" in l:                           # only the shared-filter channeliser (graded at 1e-4, never the default)
            assert "k_channelise16" in l, l
        if "soft-tolerance fma" in l:                        # only kernels instantiated for TRXSIG_SOFT_TOLERANCE (TOL = true: the last
            sym = re.search(r"\[(\w+)\]", l).group(1)       # template argument of k_demod / k_normal_quad / k_normal_chain)
            assert re.search(r"(7k_demodILi\dELb0ELi148ENS_6SmpC32ELb1EEE|k_normal_quadILi\d.*ELb1EEE|k_normal_chainILi\d.*ELb1EEE|k_demod_rxILi4ELb1EEE)", sym), l
    assert any("steering fma" in l and "k_rach_front" in l for l in rows)
    assert any("approx-form fma" in l and "k_channelise16" in l for l in rows)
    assert any("soft-tolerance fma" in l and "k_demod" in l for l in rows)
