#!/usr/bin/env python3
"""Average per-dispatch counter values per kernel from a rocprofv3 counter_collection.csv (our kernels only)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("k_gemm", "k_attn", "k_mlp", "k_stack", "k_seq", "k_x_to", "k_x_from", "k_lat", "k_randn", "k_combine", "k_to_token", "k_from_token", "k_conv", "k_block0", "k_guided")):
            continue
        k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
names = sorted({c for v in acc.values() for c in v})
# csv.writer quotes kernel names that contain commas (template arguments: "k_stack<64, false>")
out = csv.writer(sys.stdout, lineterminator="\n")
out.writerow(["kernel", "dispatches"] + names)
for k in sorted(acc):
    n = len(disp[k])
    out.writerow([k, n] + [f"{acc[k].get(c, 0.0) / n:.4g}" for c in names])
