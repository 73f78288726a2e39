#!/bin/bash
# round 6, call 6: (a) mixed-radix sizes, fp64 fields: the invariant z-pass against six components (PF_INVARIANTS=1 / 0) size by size -- which
# way the sweep should take; (b) the whole -m gpu suite on the sources of the moment, with durations
mkdir -p gpurun_out/r06
for n in 200 384 640 720 768 1000 1536; do
  for inv in 1 0; do
    PF_INVARIANTS=$inv PF_LPT_FUSE=$inv timeout 600 python3 bench.py --n $n --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 --boundary 0 --table-steps 0 > gpurun_out/r06/inv_${n}_$inv.json 2> gpurun_out/r06/inv_${n}_$inv.err
  done
done
python3 - <<'PY'
import json
for n in (200, 384, 640, 720, 768, 1000, 1536):
    row = []
    for inv in (1, 0):
        try:
            d = json.load(open(f"gpurun_out/r06/inv_{n}_{inv}.json")); row.append("%8.2f" % d["ms_per_step"])
        except Exception as e:
            row.append("failed")
    print(n, "invariants / six components: ms per step", *row)
PY
timeout 1700 python3 -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r06/gpu_suite.txt 2>&1; tail -32 gpurun_out/r06/gpu_suite.txt
