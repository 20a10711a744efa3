#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into the rows of our own kernels:
   python tools/prof_summary.py gpurun_out/prof_r1 > profiles/r01_kernel_stats.csv"""
import csv
import glob
import os
import sys

root = sys.argv[1]
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        n = row["Name"]
        if "(anonymous namespace)::k_" in n:
            # `void (anonymous namespace)::k_demod<4, false, 148, (anonymous namespace)::SmpC32, true>(TrxTables const*, ...)` -> `k_demod<4, false, 148, SmpC32, true>`
            t = n.replace("(anonymous namespace)::", "")
            t = t[5:] if t.startswith("void ") else t
            depth, cut = 0, len(t)
            for i, ch in enumerate(t):
                if ch == "<": depth += 1
                elif ch == ">": depth -= 1
                elif ch == "(" and depth == 0:
                    cut = i; break
            short = t[:cut]
            w.writerow([short, row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"],
                        row["MinNs"], row["MaxNs"], row["StdDev"]])
