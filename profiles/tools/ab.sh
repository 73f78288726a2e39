#!/bin/bash
# A/B of library builds on ONE box: bench.py once per build (csrc/Makefile VARIANT=...), per-kernel table side by side.
#   profiles/tools/ab.sh default pad w0     -> libpinfmax_hip.so, csrc/build_pad/libpinfmax_hip_pad.so, csrc/build_w0/libpinfmax_hip_w0.so
# Writes gpurun_out/ab_<name>.json; extra bench arguments through AB_ARGS.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
for v in "$@"; do
  lib=pinocchio_amd/csrc/build_$v/libpinfmax_hip_$v.so
  [ "$v" = default ] && lib=pinocchio_amd/libpinfmax_hip.so
  PINFMAX_LIB=$PWD/$lib python3 bench.py --steps ${AB_STEPS:-3} --warmup 1 --cpu-n 0 $AB_ARGS > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || tail -5 gpurun_out/ab_$v.err
done
python3 - "$@" <<'PY'
import json, sys
names = sys.argv[1:]
runs = {}
for v in names:
    try:
        runs[v] = json.load(open(f"gpurun_out/ab_{v}.json"))
    except Exception as e:
        print(v, "failed:", e)
rows = []
for v, d in runs.items():
    for k in d["kernels"]:
        if k["name"] not in rows:
            rows.append(k["name"])
print("%-26s" % "ms per launch" + "".join("%12s" % v for v in runs))
for r in rows:
    line = "%-26s" % r
    for v, d in runs.items():
        k = [x for x in d["kernels"] if x["name"] == r]
        line += "%12.3f" % (k[0]["ms_per_step"] * (d.get("kernel_table") or {}).get("steps", d["steps"]) / k[0]["launches"]) if k else "%12s" % "-"
    print(line)
print("%-26s" % "ms per step" + "".join("%12.1f" % d["ms_per_step"] for d in runs.values()))
PY
