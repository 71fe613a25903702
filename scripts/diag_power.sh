#!/bin/bash
# Power / clock readings (rocm-smi) while a 1000-step loop of the bench batch runs: is the step clock-managed by the power limit?
# Usage (GPU box): scripts/diag_power.sh > gpurun_out/power.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -v '^$' | head -30
echo "== idle above; loop running below (one reading per ~0.3 s) =="
python bench.py --steps 12000 --warmup 10 --no-cpu --no-small-batch > /tmp/bench_power.json 2>/dev/null &
BP=$!
for i in $(seq 1 600); do          # wait for the loop to start (imports, packing, graph capture take 10-30 s on a fresh box)
    w=$(rocm-smi --showpower 2>/dev/null | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$' | cut -d. -f1)
    [ "${w:-0}" -gt 400 ] && break
    sleep 0.2
done
sleep 2
for i in $(seq 1 10); do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E 'Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|hotspot)' | tr '\n' ';' | sed 's/  */ /g'
    echo
    sleep 0.3
done
wait $BP
cut -c1-220 /tmp/bench_power.json
