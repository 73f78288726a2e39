#!/bin/bash
mkdir -p gpurun_out/r05
export PF_SOLVE_BESIDE_Z=0
for rep in 1 0; do
for a in "--n 2048 --slab-of 8 --field-bytes 4" "--n 1024 --slab-of 8"; do
PF_REPLICATE_DK=$rep python3 bench.py $a --steps 2 --warmup 1 --cpu-n 0 --exact-steps 0 --table-steps 0 2>gpurun_out/r05/rep_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['steps']
ks={k['name']:k for k in d['kernels']}
print('replicated=$rep', '$a', round(d['ms_per_step'],1), d['config'].get('device_GB'), ' '.join('%s %.2f'%(n,ks[n]['ms_per_step']*st/ks[n]['launches']) for n in ('collapse_inv','zpass_c2r_hess_6to3inv','ypass_hess_3to6','xpass_hess_1to3','collapse_lpt_sources') if n in ks))" || tail -3 gpurun_out/r05/rep_err.txt
done; done
