#!/usr/bin/env python3
"""Static instruction mix per kernel from `make -C openbts-ttsou_amd/csrc asm` output, and a check that
every float v_fma/v_mac in the kernels belongs to a division / sqrt expansion: the numerical contract
forbids contracted multiply-adds anywhere else.  The one intended exception is k_rach_fast, whose
approximate steering pass uses explicit fmaf (its results are recomputed exactly before use)."""
import re
import sys
from collections import Counter

path = sys.argv[1] if len(sys.argv) > 1 else "openbts-ttsou_amd/csrc/trxsig_normal.gfx950.s"
only = sys.argv[2] if len(sys.argv) > 2 else ""
text = open(path).read()
for m in re.finditer(r"\n(_Z\w+):.*?\n(.*?)\n\.Lfunc_end", text, flags=re.S):
    name, body = m.group(1), m.group(2)
    if only and only not in name:
        continue
    ins = [l.strip() for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    ops = [i.split()[0] for i in ins]
    c = Counter()
    for o in ops:
        if o.startswith("v_"): c["valu"] += 1
        elif o.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif o.startswith("s_barrier"): c["barrier"] += 1
        elif o.startswith("s_"): c["salu"] += 1
        elif o.startswith("ds_"): c["lds"] += 1
        elif o.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
    # deliberate single-rounding multiply-adds whose product is exact (a tap component of exactly +-1):
    # emitted through inline asm with a marker comment, bit-identical to the separate mul and add
    exact = sum(1 for i in ins if "exact-product" in i)
    # ... and the fused multiply-adds of a steering pass (fma_steer, trxsig_dev.h): approximate values that only select
    # which lags are recomputed exactly
    steer = sum(1 for i in ins if "; steering" in i)
    # ... and those of a form that is approximate by construction and graded with a tolerance (the shared-filter channeliser,
    # trxsig_chan.hip: the per-carrier sums in another order)
    approx = sum(1 for i in ins if "; approx-form" in i)
    # ... and those of the tolerance-mode demodulator (fused_demod_tol, trxsig_demod.h: TRXSIG_SOFT_TOLERANCE -- soft bits within
    # 7.4e-5 of the reference's, hard bits exact; only in kernels instantiated with TOL = true)
    tol = sum(1 for i in ins if "; soft-tolerance" in i)
    fma = [i for i, o in enumerate(ops) if re.match(r"v_(fma_f|mac_f|fmac_f|mad_f|pk_fma)", o)
           and "exact-product" not in ins[i] and "; steering" not in ins[i] and "; approx-form" not in ins[i]
           and "; soft-tolerance" not in ins[i]]
    # hipcc's correctly-rounded division / sqrt expansions keep their fma's next to
    # v_div_scale / v_rcp / v_div_fmas / v_div_fixup / v_sqrt / v_rsq (f32 and f64)
    bad = 0
    for i in fma:
        lo = max(0, i - 24); hi = min(len(ops), i + 24)
        if not any(re.match(r"v_(div_scale|div_fmas|div_fixup|rcp_|sqrt_|rsq_)", o) for o in ops[lo:hi]):
            bad += 1
    kind = re.search(r"\d+(k_\w+?)ILi(\d)", name)
    label = "%s<sps=%s>" % (kind.group(1), kind.group(2)) if kind else name[:50]
    print("%-28s total %5d  %s  fma %d (outside a division: %d)%s" % (
        label, len(ops), dict(c), len(fma), bad, ("  exact-product fma %d" % exact if exact else "") +
        ("  steering fma %d" % steer if steer else "") + ("  approx-form fma %d" % approx if approx else "") +
        ("  soft-tolerance fma %d [%s]" % (tol, name) if tol else "")))
