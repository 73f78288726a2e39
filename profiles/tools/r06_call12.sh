#!/bin/bash
# round 6, call 12: fp32 fields on mixed-radix sizes -- six components (the rule since round 5) against the invariant z-pass (PF_INVARIANTS=2 forces it
# wherever it exists), now that the fp32 transforms are in the packed algebra
mkdir -p gpurun_out/r06
for n in 768 720 640 200; do
  for inv in 1 2; do
    PF_INVARIANTS=$inv PF_LPT_FUSE=$inv timeout 600 python3 bench.py --n $n --field-bytes 4 --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 --boundary 0 > gpurun_out/r06/inv32_${n}_$inv.json 2> gpurun_out/r06/inv32_${n}_$inv.err
  done
done
python3 - <<'PY'
import json
for n in (768, 720, 640, 200):
    row = []
    for inv in (1, 2):
        try:
            d = json.load(open(f"gpurun_out/r06/inv32_{n}_{inv}.json")); row.append("%8.2f" % d["ms_per_step"])
        except Exception as e:
            row.append("failed")
    print(n, "fp32 fields: six components (rule) / invariants forced: ms per step", *row)
PY
