#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_lines.py -x -q -k "2048 or several_jobs or band" 2>&1 | tail -3
for v in default nopipe default nopipe; do
  lib=pinocchio_amd/libpinfmax_hip.so; [ $v = nopipe ] && lib=pinocchio_amd/csrc/build_nopipe/libpinfmax_hip_nopipe.so
  PINFMAX_LIB=$PWD/$lib PF_SOLVE_BESIDE_Z=0 timeout 600 python3 bench.py --slab-of 8 --n 2048 --field-bytes 4 --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 --table-steps 0 > gpurun_out/r05/pipe_$v.json 2> gpurun_out/r05/pipe_$v.err
  python3 - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05/pipe_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['ms_per_step'],1), ' '.join("%s %.1f(%.2f)"%(k['name'][:12],k['ms_per_step'],k['GBps']/1000) for k in d['kernels'] if 'pass_' in k['name'] and 'zpass' not in k['name']))
PY
done
