#!/bin/bash
# usage (GPU box): scripts/prof_pmc.sh <tag> "<counters>" <bench args...>   (one --pmc pass, kernel-trace only)
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o $TAG -- python $ROOT/bench.py "$@" > $OUT/bench.log 2>&1
python $ROOT/scripts/pmc_summary.py $(find $OUT -name "*counter_collection.csv" | head -1) > $OUT/summary.txt
rm -f $(find $OUT -name "*counter_collection.csv") $(find $OUT -name "*kernel_trace.csv")
cat $OUT/summary.txt
