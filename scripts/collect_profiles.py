#!/usr/bin/env python3
"""Turn gpurun_out/prof_final_* and gpurun_out/pmc_* (scripts/prof_round.sh) into the committed summaries under
profiles/ (kernel-stats tables, PMC tables, pmc_traffic.json read by bench.py).  Usage: scripts/collect_profiles.py rNN"""
import csv, io, json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
rd = lambda p: open(os.path.join(R, p)).read().strip()
def parse_summary(text):
    """Rows of a pmc summary as dicts.  Kernel names may contain commas (template arguments); summaries written
    before the writer quoted them are still read correctly: the numeric columns are taken from the right."""
    lines = [l for l in text.splitlines() if l and not l.startswith("#")]
    names = next(csv.reader(io.StringIO(lines[0])))
    rows = []
    for l in lines[1:]:
        f = next(csv.reader(io.StringIO(l)))
        extra = len(f) - len(names)
        if extra > 0:                                    # unquoted commas inside the kernel name
            f = [",".join(f[:extra + 1])] + f[extra + 1:]
        rows.append(dict(zip(names, f)))
    return rows
def val(part, kernel, col):
    for r in parse_summary(rd(f"gpurun_out/pmc_{part}/summary.txt")):
        if kernel in r["kernel"]:
            return float(r[col])
    raise KeyError((part, kernel, col))
notes = {"b1024": "bench.py --steps 100 --warmup 10 --no-small-batch, B = 1024 clips: wave-per-sequence whole-step kernel k_seq, every launch = one persistent 10-step replay (syn_denoise_steps)",
         "b8": "bench.py --batch 8 --steps 300 --warmup 20 --no-small-batch: persistent small-batch kernel k_lat, one clip per XCD"}
for b, note in notes.items():
    subprocess.run([sys.executable, os.path.join(R, "scripts/summarize_stats.py"), os.path.join(R, f"gpurun_out/prof_final_{b}/kernel_stats.csv"),
                    os.path.join(R, f"profiles/{tag}_final_kernel_stats_{b}.txt"), f"{tag} final: {note}"], check=True)
    open(os.path.join(R, f"profiles/{tag}_final_bench_{b}.json"), "w").write(rd(f"gpurun_out/prof_final_{b}/bench.json") + "\n")
    with open(os.path.join(R, f"profiles/{tag}_final_pmc_{b}.txt"), "w") as f:
        f.write(f"# {tag} final PMC (per-dispatch averages), separate rocprofv3 --kernel-trace --pmc passes (scripts/prof_round.sh); {note.split(':')[0]} (fewer steps)\n")
        for part in ("sq", "fetch", "write"):
            f.write(rd(f"gpurun_out/pmc_{b}_{part}/summary.txt") + "\n")
        f.write("# FETCH_SIZE / WRITE_SIZE in KiB as reported; gfx950 FETCH_SIZE under-reports wide streaming reads by 2x (MI355X_MICROARCH.md, HBM section): "
                "corrected bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024.\n"
                "# SQ_WAVE_CYCLES / SQ_WAIT* / SQ_ACTIVE* in quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over 1024 SIMDs (= 32 x number of 32x32x16 bf16 MFMAs for k_seq, 16 x number of 16x16x32 ones for k_stack / k_lat); "
                "GRBM_GUI_ACTIVE summed over the 8 XCDs.\n")
traffic = lambda b, k: int((2 * val(f"{b}_fetch", k, "FETCH_SIZE") + val(f"{b}_write", k, "WRITE_SIZE")) * 1024)
json.dump({"batch": 1024, "layer_mode": 0, "steps_per_launch": 10, "source": f"profiles/{tag}_final_pmc_b1024.txt",
           "note": "bytes per launch (a k_seq launch = 10 steps) = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 (gfx950 FETCH_SIZE correction); fabric-side traffic, "
                   "includes the weight tape pulled by each of the 8 XCD L2s (served by the Infinity Cache)",
           "hbm_bytes_per_launch": {"k_seq": traffic("b1024", "k_seq")},
           "small_batch": {"batch": 8, "source": f"profiles/{tag}_final_pmc_b8.txt", "k_lat": traffic("b8", "k_lat"),
                           "note": "each of the 8 XCD L2s pulls the 38 MB weight set once per step (304 MB) + activations"}},
          open(os.path.join(R, "profiles/pmc_traffic.json"), "w"), indent=1)
print(open(os.path.join(R, "profiles/pmc_traffic.json")).read())
