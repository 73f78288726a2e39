#!/bin/bash
# A/B of two versions of pf_collapse_core.h on one box (scratch/ab_core_{old,new}.txt), with the grid of the collapse kernel varied
line() { timeout 300 python bench.py --steps 3 --warmup 1 --cpu-n 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['name']: round(x['ms_per_step'],1) for x in d['kernels']}
print('$1', round(d['ms_per_step'],1), k['collapse'], round(d['roofline']['avg_ms'],3))"; }
for v in new old; do
  cp scratch/ab_core_$v.txt pinocchio_amd/csrc/pf_collapse_core.h
  make -C pinocchio_amd/csrc -s 2>&1 | grep -v warning | tail -2
  for w in 8 6 12 5 10; do PF_COLLAPSE_WG_PER_CU=$w line "$v wg/cu=$w"; done
done
cp scratch/ab_core_new.txt pinocchio_amd/csrc/pf_collapse_core.h
