#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/prof_stats.sh <tag> <bench args...>
# rocprofv3 kernel-trace + stats of bench.py; writes CSVs under gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python $ROOT/bench.py "$@" > $OUT/bench.log 2>&1
grep "^{\"metric\"" $OUT/bench.log | tail -1 > $OUT/bench.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -f $(find $OUT -name "*kernel_trace.csv")   # large; the stats table is what gets committed
head -20 $OUT/kernel_stats.csv
