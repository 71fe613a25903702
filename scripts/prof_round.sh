#!/bin/bash
# One profiling round on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes for the headline
# batch (whole-step kernel) and for a small batch (persistent small-batch kernel).  Output: gpurun_out/prof_*, pmc_*.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
T="timeout 240"
$T scripts/prof_stats.sh final_b1024 --steps 100 --warmup 10 --cpu-seconds 6 --no-small-batch --no-extras > /dev/null 2>&1
$T scripts/prof_stats.sh final_b8 --batch 8 --steps 300 --warmup 20 --no-cpu --no-small-batch --no-extras > /dev/null 2>&1
SQ="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
$T scripts/prof_pmc.sh b1024_sq "$SQ" --steps 10 --warmup 10 --no-cpu --no-small-batch --no-extras > /dev/null 2>&1
$T scripts/prof_pmc.sh b1024_fetch "FETCH_SIZE" --steps 10 --warmup 10 --no-cpu --no-small-batch --no-extras > /dev/null 2>&1
$T scripts/prof_pmc.sh b1024_write "TCC_HIT_sum TCC_MISS_sum WRITE_SIZE" --steps 10 --warmup 10 --no-cpu --no-small-batch --no-extras > /dev/null 2>&1
$T scripts/prof_pmc.sh b8_sq "$SQ" --batch 8 --steps 20 --warmup 5 --no-cpu --no-small-batch --no-extras > /dev/null 2>&1
$T scripts/prof_pmc.sh b8_fetch "FETCH_SIZE" --batch 8 --steps 20 --warmup 5 --no-cpu --no-small-batch --no-extras > /dev/null 2>&1
$T scripts/prof_pmc.sh b8_write "TCC_HIT_sum TCC_MISS_sum WRITE_SIZE" --batch 8 --steps 20 --warmup 5 --no-cpu --no-small-batch --no-extras > /dev/null 2>&1
for d in gpurun_out/prof_final_b1024 gpurun_out/prof_final_b8; do echo "== $d"; cat $d/bench.json | cut -c1-300; head -4 $d/kernel_stats.csv | cut -c1-200; done
for d in gpurun_out/pmc_*; do echo "== $d"; cat $d/summary.txt | grep -E "kernel|k_stack|k_seq|k_lat"; done
