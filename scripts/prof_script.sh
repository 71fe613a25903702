#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/prof_script.sh <tag> <script.py> [args...]
# rocprofv3 kernel-trace + stats of any script of this repo; writes gpurun_out/prof_<tag>/kernel_stats.csv + run.log
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python $SCRIPT "$@" > $OUT/run.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -f $(find $OUT -name "*kernel_trace.csv")
head -12 $OUT/kernel_stats.csv
