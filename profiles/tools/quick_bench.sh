#!/bin/bash
# quick look at a build ON THE GPU BOX: a short default-order bench (no CPU baseline, no exact-libm run) and its per-kernel table
tag=${1:-quick}; shift
mkdir -p gpurun_out/r04
timeout 600 python3 bench.py --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 "$@" > gpurun_out/r04/bench_$tag.json 2> gpurun_out/r04/bench_$tag.err
python3 - "$tag" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r04/bench_{sys.argv[1]}.json"))
print("ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("kernel", "avg_ms", "achieved", "frac")})
print("in-line pass ms_per_step", (d.get("kernel_table") or {}).get("ms_per_step"))
for k in d.get("kernels", []):
    print("%-26s %3d launches %8.2f ms/step %7.0f GB/s  %s" % (k["name"], k["launches"], k["ms_per_step"], k["GBps"], k["symbol"][:60]))
PY
