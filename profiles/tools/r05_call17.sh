#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_lines.py -x -q -k "invariant" 2>&1 | tail -3
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "invariant_zpass" 2>&1 | tail -3
AB_ARGS="--n 768" AB_STEPS=2 bash profiles/tools/ab.sh default ziw0 2>&1 | tail -16
AB_ARGS="--n 200" AB_STEPS=5 bash profiles/tools/ab.sh default ziw0 2>&1 | tail -16
for inv in 1 0; do
PF_INVARIANTS=$inv PF_LPT_FUSE=$inv timeout 600 python3 bench.py --n 768 --field-bytes 4 --steps 2 --warmup 1 --cpu-n 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=(d.get('kernel_table') or {}).get('steps', d['steps']); print('768 fp32 inv=$inv', round(d['ms_per_step'],1), ' '.join('%s %.2f'%(k['name'],k['ms_per_step']*st/k['launches']) for k in d['kernels']))"
done
