#!/bin/bash
# Fabric-side bytes per kernel of ONE training step (GPU box): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over
# scripts/prof_train_eager.py (2 warm-up + 3 counted eager steps), summed per kernel and step by scripts/pmc_train_summary.py.
# FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md "HBM"); Infinity-Cache hits are counted, not excluded.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o run -- python $ROOT/scripts/prof_train_eager.py ${1:-32} > $OUT/$C.log 2>&1
done
python $ROOT/scripts/pmc_train_summary.py $(find $OUT/FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/WRITE_SIZE -name "*counter_collection.csv" | head -1) 5 3 > $OUT/summary.txt
find $OUT -name "*.csv" -delete
cat $OUT/summary.txt
