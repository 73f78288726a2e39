#!/bin/bash
# Compute side of ONE rank of the multi-GPU configurations, each on one GPU with the exchange short-circuited (bench.py --slab-of,
# pf_set_loopback_exchange): slab_matrix.sh <tag>  ->  gpurun_out/<tag>_slab_<n>_p<P>[_rep<0|1>][_fp32].json and a table.
# Not a scaling measurement: the all-to-alls are not in it.  Run from the repo root on the GPU box.
tag=${1:-r03}
mkdir -p gpurun_out
run() {  # run <name> <env assignment or -> <bench args...>
  name=$1; envs=$2; shift 2
  if [ "$envs" = "-" ]; then python3 bench.py "$@" > gpurun_out/${tag}_slab_$name.json 2> gpurun_out/${tag}_slab_$name.err
  else env $envs python3 bench.py "$@" > gpurun_out/${tag}_slab_$name.json 2> gpurun_out/${tag}_slab_$name.err; fi
  [ -s gpurun_out/${tag}_slab_$name.json ] || tail -2 gpurun_out/${tag}_slab_$name.err
}
for P in 2 4 8; do
  for rep in 0 1; do run 1024_p${P}_rep$rep PF_REPLICATE_DK=$rep --slab-of $P --n 1024 --steps 3 --warmup 1; done
done
run 2048_p8_fp32 - --slab-of 8 --n 2048 --field-bytes 4 --steps 3 --warmup 1
run 1024_p8_fp32 - --slab-of 8 --n 1024 --field-bytes 4 --steps 3 --warmup 1
python3 - "$tag" <<'PY'
import glob, json, sys
tag = sys.argv[1]
print("%-22s %10s %14s %16s %8s" % ("slab", "ms/step", "cells/s/rank", "box cells/s (*)", "GB"))
for f in sorted(glob.glob(f"gpurun_out/{tag}_slab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e); continue
    print("%-22s %10.1f %14.3e %16.3e %8.1f" % (f.split("_slab_")[1][:-5], d["ms_per_step"], d["value"], d["box_cells_per_s_if_the_exchanges_hide"], d["config"]["device_GB"]))
print("(*) if every all-to-all hid behind the kernels: an upper bound, not a measurement of P GPUs")
PY
