#!/bin/bash
# A/B of two versions of pf_cell_kernels.hip on one box: new, old, new, old
line() { timeout 300 python bench.py --steps 3 --warmup 1 --cpu-n 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['name']: round(x['ms_per_step'],1) for x in d['kernels']}
print('$1', round(d['ms_per_step'],1), k['xpass_hess_1to3'], k['ypass_hess_3to6'], k['zpass_c2r_hess_6'], k['collapse'], round(d['roofline']['avg_ms'],3))"; }
for v in new old new old; do
  cp scratch/ab_cell_$v.txt pinocchio_amd/csrc/pf_cell_kernels.hip
  make -C pinocchio_amd/csrc -s 2>&1 | grep -v warning | tail -2
  line $v
done
