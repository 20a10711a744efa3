"""Numerical-contract audit on the generated gfx950 ISA (no GPU needed: hipcc cross-compiles): outside
hipcc's correctly-rounded division / sqrt expansions no kernel may contain a fused multiply-add --
except k_rach_fast, whose approximate steering pass uses explicit fmaf and recomputes everything it
hands on exactly, and the marked exact-product FMAs of the midamble correlators (a tap component of
exactly +-1: single rounding == separate mul and add; tools/asm_stats.py counts them apart)."""
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
    for f in ("trxsig_normal", "trxsig_fused", "trxsig_rach", "trxsig_eq", "trxsig_tx", "trxsig_fec"):
        f += ".gfx950.s"
        out += subprocess.check_output(["python3", os.path.join(ROOT, "tools", "asm_stats.py"),
                                        os.path.join(ROOT, "openbts-ttsou_amd", "csrc", f)], text=True)
    rows = [l for l in out.splitlines() if "outside a division" in l]
    assert len(rows) >= 23 and any("k_fec_viterbi" in l for l in rows)
    for l in rows:
        n = int(re.search(r"outside a division: (\d+)", l).group(1))
        if "k_rach_fast" in l:
            assert n > 0            # the explicit fmaf of the approximate pass
        else:
            assert n == 0, l
