#!/bin/bash
export PF_SOLVE_BESIDE_Z=0
for a in "--n 1024" "--n 1024 --slab-of 2" "--n 1024 --slab-of 8" "--n 1024 --field-bytes 4" "--n 2048 --slab-of 8 --field-bytes 4" "--n 1024 --slab-of 8 --field-bytes 4" "--n 512"; do
python3 bench.py $a --steps 2 --warmup 1 --cpu-n 0 --exact-steps 0 --table-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['steps']
ks={k['name']:k for k in d['kernels']}
cells=d['config'].get('cells_per_rank') or 0
print('$a', round(d['ms_per_step'],1), ' '.join('%s %.2f'%(n,ks[n]['ms_per_step']*st/ks[n]['launches']) for n in ('collapse_inv','zpass_c2r_hess_6to3inv','collapse_lpt_sources') if n in ks), {k:v for k,v in d['config'].items() if k in ('slab_of','grid','field_bytes')})"
done
