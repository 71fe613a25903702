#!/bin/bash
# On the GPU box: time every library under syntalker_amd/csrc/variants/ (or the ones named) with scripts/diag_seq_quick.py, twice (clock drift).
# Usage: scripts/run_variants.sh [B] [name ...]
B=${1:-1024}; shift || true
V=syntalker_amd/csrc/variants
names=("$@"); if [ ${#names[@]} -eq 0 ]; then names=($(ls $V | sed -n 's/^libsyn_\(.*\)\.so$/\1/p')); fi
for round in 1 2; do
  for n in "${names[@]}"; do
    SYN_HIP_LIB=$PWD/$V/libsyn_$n.so timeout 300 python scripts/diag_seq_quick.py $B $n 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
