#!/bin/bash
# z-pass persistent-kernel experiment: parity subset, then bench at a few settings
export PF_ZPASS_PERSIST=3
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "transforms or second_derivatives or full_path_vs_oracle or golden or fp32" 2>&1 | tail -3
for v in 0 3 6; do
  export PF_ZPASS_PERSIST=$v
  echo "== PF_ZPASS_PERSIST=$v"
  timeout 600 python bench.py --steps 2 --warmup 1 --cpu-n 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'],1))
for k in d['kernels']:
    if k['launches']: print('  ', k['name'], k['launches'], round(k['ms_per_step'],1), 'ms', round(k['GBps'],0), 'GB/s')
"
done
