#!/bin/bash
for args in "--n 256 --no-lpt" "--n 256" "--n 512" "--n 1024 --field-bytes 4" "--n 1024 --no-lpt"; do
  echo "== $args"
  timeout 600 python bench.py $args --steps 3 --warmup 1 --cpu-n 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.2f  cells/s %.3e  contract_frac %.3f  device_GB %.1f' % (d['ms_per_step'], d['value'], d['path_roofline']['frac_of_hbm_peak'], d['config']['device_GB']))
"
done
