#!/bin/bash
# Build A/B variants of libsyn_hip.so in parallel: scripts/seq_variants.sh name1="-DFOO=1 -DBAR" name2="" ...
# -> syntalker_amd/csrc/variants/libsyn_<name>.so (git-ignored, travels with gpurun); run with SYN_HIP_LIB=<path>.
set -uo pipefail
SRC=/root/repo/syntalker_amd/csrc
mkdir -p $SRC/variants
pids=()
for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    ( cd $SRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=262144 -Wno-unused-result -Wno-unused-value \
        -I../../include -shared -fPIC $flags syn_kernels.hip -o variants/libsyn_$name.so -Rpass-analysis=kernel-resource-usage > /tmp/variant_$name.log 2>&1
      sp=$(grep -A12 "Function Name: .*k_seq" /tmp/variant_$name.log | grep "VGPRs Spill" | sed 's/.*VGPRs Spill: \([0-9]*\).*/\1/' | tr '\n' ' ')
      if [ -f variants/libsyn_$name.so ]; then echo "built $name [$flags] k_seq spills: $sp"; else echo "FAILED $name"; grep -m5 "error" /tmp/variant_$name.log; fi ) &
    pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
