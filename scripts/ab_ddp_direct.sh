#!/bin/bash
# A/B of the captured DDP step's gradient path on one box (1-rank RCCL): PyTorch DDP as it is | + averaging comm hook | + gradients written
# straight into the bucket views, next to the plain captured step.   bash scripts/ab_ddp_direct.sh
cd "$(dirname "$0")/.."
run() { timeout 300 python bench.py --mode train "$@" --steps 200 --warmup 10 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms, loss', d['loss'])"; }
for rep in 1 2; do
  echo -n "plain captured step:                 "; run
  echo -n "DDP, built-in reducer path:          "; SYN_DDP_AVG_HOOK=0 run --force-ddp
  echo -n "DDP + comm hook (per-param copies):  "; SYN_DDP_AVG_HOOK=1 SYN_DDP_DIRECT_GRADS=0 run --force-ddp
  echo -n "DDP + comm hook + direct gradients:  "; SYN_DDP_AVG_HOOK=1 SYN_DDP_DIRECT_GRADS=1 run --force-ddp
done
