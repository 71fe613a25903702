#!/bin/bash
# A/B of the split-operand training convolutions' cross products: per configuration "fwd,dgrad,wgrad"
# (bit 0: A_lo . B_hi, bit 1: A_hi . B_lo; 3 = both) the worst per-tensor gradient error of the two training goldens and the
# captured step's time.   bash scripts/ab_conv_terms.sh > gpurun_out/r04/ab_conv_terms.txt
cd "$(dirname "$0")/.."
for cfg in 3,3,3 2,3,3 1,3,3 3,2,3 3,1,3 3,3,2 3,3,1 2,2,2 1,1,1 0,0,0 "$@"; do
  echo "== SYN_CONV_TERMS=$cfg"
  SYN_CONV_TERMS=$cfg timeout 300 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "test_training_loss_and_gradients or test_train_mode_loss_gradients" 2>&1 \
      | grep -E "worst|passed|failed|AssertionError|assert " | head -8
  SYN_CONV_TERMS=$cfg timeout 300 python bench.py --mode train --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   captured step', d['ms_per_step'], 'ms, loss', d['loss'])"
done
