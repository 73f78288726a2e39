#!/bin/bash
# replicated A/B: ab_rep.sh REPS "name:VARS" ...  -> interleaved runs, per-class ms per launch as mean (min..max)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
reps=$1; shift
for r in $(seq 1 $reps); do
  for spec in "$@"; do
    name=${spec%%:*}; vars=${spec#*:}
    env $vars python3 bench.py --steps ${AB_STEPS:-2} --warmup 1 --cpu-n 0 --exact-steps 0 $AB_ARGS > gpurun_out/abr_${name}_$r.json 2> gpurun_out/abr_${name}_$r.err || tail -3 gpurun_out/abr_${name}_$r.err
  done
done
python3 - $reps "$@" <<'PY'
import json, sys
reps = int(sys.argv[1]); names = [s.split(":")[0] for s in sys.argv[2:]]
runs = {v: [json.load(open(f"gpurun_out/abr_{v}_{r}.json")) for r in range(1, reps + 1)] for v in names}
rows = []
for v, ds in runs.items():
    for k in ds[0]["kernels"]:
        if k["name"] not in rows: rows.append(k["name"])
print("%-24s" % "ms per launch" + "".join("%22s" % v for v in runs))
for r in rows:
    line = "%-24s" % r
    for v, ds in runs.items():
        vals = []
        for d in ds:
            k = [x for x in d["kernels"] if x["name"] == r]
            if k: vals.append(k[0]["ms_per_step"] * d["steps"] / k[0]["launches"])
        line += "%8.2f (%5.2f..%5.2f)" % (sum(vals) / len(vals), min(vals), max(vals)) if vals else "%22s" % "-"
    print(line)
line = "%-24s" % "ms per step"
for v, ds in runs.items():
    vals = [d["ms_per_step"] for d in ds]
    line += "%8.1f (%5.0f..%5.0f)" % (sum(vals) / len(vals), min(vals), max(vals))
print(line)
PY
