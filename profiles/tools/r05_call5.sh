#!/bin/bash
# round 5: A/B of the solve (branchless sqrt, exp as one asm statement, next cell's inputs in flight), config-5 slab with the slimmer filter,
# and the Fmax contract of the new default at 128^3 / 256^3
mkdir -p gpurun_out/r05
AB_ARGS="--exact-steps 0 --table-steps 2" AB_STEPS=3 bash profiles/tools/ab.sh default nopf old > gpurun_out/r05/ab_solve.txt 2>&1
cat gpurun_out/r05/ab_solve.txt
timeout 600 python3 profiles/tools/fmax_contract.py 128 > gpurun_out/r05/contract128.txt 2>&1; cat gpurun_out/r05/contract128.txt
timeout 900 python3 profiles/tools/fmax_contract.py 256 > gpurun_out/r05/contract256.txt 2>&1; cat gpurun_out/r05/contract256.txt
PF_SOLVE_BESIDE_Z=0 timeout 600 python3 bench.py --slab-of 8 --n 2048 --field-bytes 4 --steps 3 --warmup 1 > gpurun_out/r05/s16b_slab2048_inline.json 2> gpurun_out/r05/s16b_slab2048_inline.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/s16b_slab2048_inline.json').read().strip().splitlines()[-1])
print(d['ms_per_step'])
for k in d['kernels']: print("  %-26s %3d %8.2f ms/step %7.0f GB/s %s"%(k['name'],k['launches'],k['ms_per_step'],k['GBps'],k['symbol'][:50]))
PY
