#!/bin/bash
# Build libsyn_hip.so for gfx950 in-tree (the .so is git-ignored but travels with gpurun snapshots).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -I../../include -shared -fPIC "$@" syn_kernels.hip -o libsyn_hip.so
echo "built $(pwd)/libsyn_hip.so"
