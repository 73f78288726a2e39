#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_lines.py -x -q -k "packed or 2048 or 1024-" > gpurun_out/r05/lines_pk8.log 2>&1
tail -5 gpurun_out/r05/lines_pk8.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "fp32" > gpurun_out/r05/parity_fp32.log 2>&1
tail -5 gpurun_out/r05/parity_fp32.log
timeout 600 python3 bench.py --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 --field-bytes 4 > gpurun_out/r05/pk8_fp32_1024.json 2> gpurun_out/r05/pk8_fp32_1024.err
tail -3 gpurun_out/r05/pk8_fp32_1024.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/pk8_fp32_1024.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('result_check',{}).get('matches_single_gpu_golden'))
for k in d['kernels']: print("  %-26s %3d %8.2f ms/step %7.0f GB/s %s"%(k['name'],k['launches'],k['ms_per_step'],k['GBps'],k['symbol'][:50]))
PY
