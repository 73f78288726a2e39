#!/bin/bash
export PF_SOLVE_BESIDE_Z=0
for cp in 0 1000000; do
for a in "--n 2048 --slab-of 8 --field-bytes 4" "--n 1024 --slab-of 8"; do
PF_BENCH_LOOPBACK_COPIES=$cp python3 bench.py $a --steps 2 --warmup 1 --cpu-n 0 --exact-steps 0 --table-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['steps']
ks={k['name']:k for k in d['kernels']}
print('copies=$cp', '$a', round(d['ms_per_step'],1), ' '.join('%s %.2f'%(n,ks[n]['ms_per_step']*st/ks[n]['launches']) for n in ('collapse_inv','zpass_c2r_hess_6to3inv','ypass_hess_3to6','collapse_lpt_sources') if n in ks))"
done; done
