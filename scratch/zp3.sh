#!/bin/bash
for v in 24 48; do PF_ZPASS_PERSIST=$v timeout 300 python scratch/zmicro.py; done
ZN=512 PF_ZPASS_PERSIST=0 timeout 300 python scratch/zmicro.py
ZN=512 PF_ZPASS_PERSIST=24 timeout 300 python scratch/zmicro.py
ZN=256 PF_ZPASS_PERSIST=0 timeout 300 python scratch/zmicro.py
ZN=256 PF_ZPASS_PERSIST=24 timeout 300 python scratch/zmicro.py
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-n 0 > gpurun_out/bench_zp.json; python -c "
import json
d=json.loads(open('gpurun_out/bench_zp.json').read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'],1), d['value'])
for k in d['kernels']:
    if k['launches']: print('  ', k['name'], k['launches'], round(k['ms_per_step'],1), 'ms', round(k['GBps'],0), 'GB/s')
"
