#!/bin/bash
# Training-side profiles of a round (GPU box, via gpurun): rocprofv3 kernel stats of bench.py --mode train, the torch-profiler kernel split of one step,
# the default bench line and the batch sweep.  Output under gpurun_out/; scripts/collect_profiles.py does not touch these - copy them by hand.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r06
timeout 300 scripts/prof_stats.sh train --mode train --steps 100 --warmup 10 > /dev/null 2>&1
python scripts/summarize_stats.py gpurun_out/prof_train/kernel_stats.csv gpurun_out/r06/train_kernel_stats.txt "r06: bench.py --mode train --steps 100 --warmup 10 (captured step replayed 110 times + 3 eager warm-up + 3 eager split steps: 116 steps; divide calls by 116)"
cp gpurun_out/prof_train/bench.json gpurun_out/r06/bench_train.json
{ echo "# scripts/prof_train_steady.py 32 90 (torch.profiler, device activities, 5 eager steps after warm-up; round 6 HEAD): device time per kernel of one training step at 32 clips"; python scripts/prof_train_steady.py 32 90 2>&1 | grep -v -i "warn\|amdgpu"; } > gpurun_out/r06/train_step_split.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
{ echo "# scripts/diag_batch_sweep.py, round 6 (one MI355X box): steady-state DDPM step, library's kernel choice and slicing"; python scripts/diag_batch_sweep.py 2>&1 | grep -v amdgpu; } > gpurun_out/r06/diag_batch_sweep.txt
python scripts/diag_stack_train.py 32 2>&1 | grep -v amdgpu > gpurun_out/r06/diag_stack_train.txt
tail -3 gpurun_out/r06/train_step_split.txt; head -c 300 gpurun_out/r06/bench_default.json; echo; tail -5 gpurun_out/r06/diag_batch_sweep.txt
