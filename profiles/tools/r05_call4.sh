#!/bin/bash
# round 5: first contact of the sixteen-point 2048 kernel: algebra + line tests, then the config-5 slab bench
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_lines.py -x -q -k "packed or 2048 or 1024-" > gpurun_out/r05/lines16.log 2>&1
tail -15 gpurun_out/r05/lines16.log
PF_SOLVE_BESIDE_Z=0 timeout 600 python3 bench.py --slab-of 8 --n 2048 --field-bytes 4 --steps 3 --warmup 1 > gpurun_out/r05/s16_slab2048_inline.json 2> gpurun_out/r05/s16_slab2048_inline.err
tail -3 gpurun_out/r05/s16_slab2048_inline.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/s16_slab2048_inline.json').read().strip().splitlines()[-1])
print(d['ms_per_step'])
for k in d['kernels']: print("  %-26s %3d %8.2f ms/step %7.0f GB/s %s"%(k['name'],k['launches'],k['ms_per_step'],k['GBps'],k['symbol'][:50]))
PY
