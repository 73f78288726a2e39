#!/bin/bash
# A/B of two versions of pf_collapse_core.h on one box (scratch/ab_core_{old,new}.txt, not kept in the repository)
line() { timeout 300 python bench.py --steps 3 --warmup 1 --cpu-n 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['name']: round(x['ms_per_step'],1) for x in d['kernels']}
print('$1', round(d['ms_per_step'],1), k['collapse'], round(d['roofline']['avg_ms'],3))"; }
for v in new old new old; do
  cp scratch/ab_core_$v.txt pinocchio_amd/csrc/pf_collapse_core.h
  make -C pinocchio_amd/csrc -s 2>&1 | grep -v warning | tail -2
  line "$v"
done
cp scratch/ab_core_new.txt pinocchio_amd/csrc/pf_collapse_core.h
make -C pinocchio_amd/csrc -s 2>&1 | grep -v warning | tail -2
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "elementary or full_path or hmf or golden or kat" 2>&1 | tail -3
