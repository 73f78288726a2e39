#!/bin/bash
for rep in 1 2; do
for z in 1 0; do
for a in "--n 1024" "--n 1024 --field-bytes 4"; do
PF_SOLVE_BESIDE_Z=$z python3 bench.py $a --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 --table-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('beside=$z', '$a', round(d['ms_per_step'],1), 'roofline', d['roofline']['kernel'][:30], round(d['roofline']['frac'],3), round(d['roofline']['ms_per_step'],1))"
done; done; done
