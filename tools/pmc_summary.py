#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from the csv files under a tools/pmc.sh output directory."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "")
        k = re.sub(r"\(.*", "", k).replace("void ", "")
        if not k.startswith("k_"):
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-24s avg %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
