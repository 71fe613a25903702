#!/usr/bin/env python3
"""Trim a rocprofv3 kernel_stats.csv to a committed summary: our kernels in full + the top foreign kernels."""
import csv
import sys

src, dst, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
rows = list(csv.DictReader(open(src)))
OURS = ("k_stack", "k_seq", "k_x_to_fragment", "k_x_from_fragment", "k_lat", "k_gemm", "k_attn", "k_mlp", "k_pack", "k_to_token", "k_from_token", "k_randn", "k_combine", "k_axpby",
        "k_conv", "k_block0", "k_guided", "k_quantize", "k_codes", "k_vq_pose_in", "k_ln_", "k_gelu_", "k_colsum", "k_step_advance",
        "k_bn_", "k_opt_", "k_part_sums", "k_linear_", "k_embedding", "k_masked", "k_rotary", "k_mfma_rate", "k_cond_")
ours = [r for r in rows if any(t in r["Name"] for t in OURS) and "at::native" not in r["Name"]]
other = [r for r in rows if r not in ours][:8]
with open(dst, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary ({note})\n")
    f.write("# columns: calls, total_ms, avg_us, pct_of_all_gpu_time, min_us, max_us, kernel\n")
    f.write("## syntalker_amd kernels\n")
    for r in ours:
        n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        f.write(f"{int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} "
                f"{float(r['Percentage']):6.2f} {float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f}  {n[:90]}\n")
    f.write("## largest other kernels (PyTorch-ROCm ops of the per-clip conditioning, fills, copies)\n")
    for r in other:
        f.write(f"{int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} "
                f"{float(r['Percentage']):6.2f} {float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f}  {r['Name'][:90]}\n")
