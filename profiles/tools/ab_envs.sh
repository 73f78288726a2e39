#!/bin/bash
# A/B of environment settings on one box: ab_envs.sh "name1:VAR=a VAR2=b" "name2:VAR=c" ...   (name "base:" = no variables)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
names=()
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  names+=("$name")
  env $vars python3 bench.py --steps ${AB_STEPS:-3} --warmup 1 --cpu-n 0 --exact-steps 0 $AB_ARGS > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || tail -3 gpurun_out/ab_$name.err
done
python3 - "${names[@]}" <<'PY'
import json, sys
names = sys.argv[1:]
runs = {v: json.load(open(f"gpurun_out/ab_{v}.json")) for v in names}
rows = []
for v, d in runs.items():
    for k in d["kernels"]:
        if k["name"] not in rows: rows.append(k["name"])
print("%-26s" % "ms per launch" + "".join("%11s" % v for v in runs))
for r in rows:
    line = "%-26s" % r
    for v, d in runs.items():
        k = [x for x in d["kernels"] if x["name"] == r]
        line += "%11.3f" % (k[0]["ms_per_step"] * d["steps"] / k[0]["launches"]) if k else "%11s" % "-"
    print(line)
print("%-26s" % "ms per step" + "".join("%11.1f" % d["ms_per_step"] for d in runs.values()))
PY
