#!/usr/bin/env python3
"""Per kernel and training step: launches, fabric-side MB read (2 x FETCH_SIZE KiB, gfx950) and written (WRITE_SIZE KiB), from two rocprofv3
counter_collection.csv files over the same `steps_total` eager steps, of which the last `steps_counted` are summed (dispatch order)."""
import csv
import sys
from collections import defaultdict


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def short(k):
    return k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]


fetch, write = load(sys.argv[1]), load(sys.argv[2])
total, counted = int(sys.argv[3]), int(sys.argv[4])
out = defaultdict(lambda: [0, 0.0, 0.0])
for rows, col in ((fetch, 1), (write, 2)):
    # a step ends with its last k_opt_adam launch; the backward chain kernel runs once per step: count from the (total - counted + 1)-th of those on
    marks = [i for i, (_, k, _) in enumerate(rows) if "k_stack_train_bwd" in k]
    assert len(marks) == total, (len(marks), total)
    first = max(i for i, (_, k, _) in enumerate(rows[:marks[total - counted]]) if "k_opt_adam" in k) + 1
    for _, k, v in rows[first:]:
        e = out[short(k)]
        if col == 1:
            e[0] += 1
        e[col] += v
scale_f, scale_w = 2 * 1024 / 1e6 / counted, 1024 / 1e6 / counted
tab = sorted(((k, e[0] / counted, e[1] * scale_f, e[2] * scale_w) for k, e in out.items()), key=lambda r: -(r[2] + r[3]))
print(f"# fabric-side traffic of one eager training step (mean of the last {counted} of {total}); MB read = 2 x FETCH_SIZE KiB (gfx950 correction), MB written = WRITE_SIZE KiB")
print(f"total: {sum(r[2] for r in tab):9.1f} MB read {sum(r[3] for r in tab):9.1f} MB written")
print(f"{'launches':>8} {'MB read':>9} {'MB written':>10}  kernel")
for k, n, r, w in tab[:60]:
    print(f"{n:8.1f} {r:9.1f} {w:10.1f}  {k}")
