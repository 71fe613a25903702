#!/bin/bash
# Register / scratch statistics of one kernel of syn_kernels.hip (default: the small-batch kernel), plus where
# its scratch traffic sits relative to the s_memtime debug stamps.  Usage: scripts/isa_stats.sh [name-pattern] [-Dflags]
set -euo pipefail
PAT=${1:-lat5k_lat}; shift || true
SRC=/root/repo/syntalker_amd/csrc
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -I/root/repo/include "$@" \
    -S --cuda-device-only $SRC/syn_kernels.hip -o /tmp/isa/k.s
awk "/\.name:.*$PAT/,/\.wavefront_size/" /tmp/isa/k.s | grep -E "\.name|vgpr_count|spill|private_segment_fixed|group_segment_fixed"
awk "/^_Z.*$PAT.*:/,/s_endpgm/" /tmp/isa/k.s > /tmp/isa/kernel.s
awk '/s_memtime/{n++} /scratch_store/{st[n]++} /scratch_load/{ld[n]++} END{for(i=0;i<=n;i++) if (st[i]+ld[i]>0) printf "stamp-region %d: %d scratch stores, %d loads; ", i, st[i], ld[i]; print ""}' /tmp/isa/kernel.s
