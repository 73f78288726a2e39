#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_lines.py -x -q -k "invariant" 2>&1 | tail -4
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "invariant_zpass or general or example_size_200 or grid_200 or lpt" 2>&1 | tail -4
for n in 768 200; do
  st=2; [ $n = 200 ] && st=5
  for inv in 1 0; do
    PF_INVARIANTS=$inv PF_LPT_FUSE=$inv timeout 600 python3 bench.py --n $n --steps $st --warmup 1 --cpu-n 0 > gpurun_out/r05/inv_${n}_$inv.json 2> gpurun_out/r05/inv_${n}_$inv.err
    python3 - $n $inv <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05/inv_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
st=(d.get('kernel_table') or {}).get('steps', d['steps'])
print(sys.argv[1], 'inv' if sys.argv[2]=='1' else 'six', round(d['ms_per_step'],2), d.get('result_check',{}).get('ok'), ' '.join("%s %.3f"%(k['name'],k['ms_per_step']*st/k['launches']/ (st)) for k in d['kernels']))
PY
  done
done
PF_INVARIANTS=1 timeout 600 python3 bench.py --n 768 --field-bytes 4 --steps 2 --warmup 1 --cpu-n 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('768 fp32', round(d['ms_per_step'],1), ' '.join('%s %.2f'%(k['name'],k['ms_per_step']/k['launches']) for k in d['kernels']))"
