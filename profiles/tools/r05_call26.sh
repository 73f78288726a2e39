#!/bin/bash
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "invariant or general or example_size_200 or grid_200 or lpt" 2>&1 | tail -5
timeout 600 python3 bench.py --n 1536 --field-bytes 4 --steps 2 --warmup 1 --cpu-n 0 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=(d.get('kernel_table') or {}).get('steps', d['steps']); print('1536 fp32', round(d['ms_per_step'],1), d['config'].get('device_GB'), ' '.join('%s %.2f'%(k['name'],k['ms_per_step']*st/k['launches']) for k in d['kernels']))"
