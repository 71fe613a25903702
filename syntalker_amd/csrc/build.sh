#!/bin/bash
# Build libsyn_hip.so for gfx950 in-tree (the .so is git-ignored but travels with gpurun snapshots).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -pragma-unroll-threshold: k_seq's GEMM pieces are straight-line code by design (every weight fragment's position in the
# LDS ring is a compile-time constant); the default limit of 16 k instructions per unrolled loop refuses the larger ones.
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=262144 -Wno-unused-result -Wno-unused-value -I../../include -shared -fPIC "$@" syn_kernels.hip -o libsyn_hip.so
echo "built $(pwd)/libsyn_hip.so"
