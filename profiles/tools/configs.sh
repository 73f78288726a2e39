#!/bin/bash
# the single-GPU configurations of BASELINE.json through bench.py (run from the repo root on the GPU box)
mkdir -p gpurun_out
run() { name=$1; shift; python3 bench.py --cpu-n 0 --exact-steps 0 "$@" > gpurun_out/cfg_$name.json 2> gpurun_out/cfg_$name.err || tail -3 gpurun_out/cfg_$name.err; }
run 256_fmax --n 256 --no-lpt --steps 20 --warmup 3
run 256_full --n 256 --steps 20 --warmup 3
run 512_full --n 512 --steps 10 --warmup 2
run 1024_fmax --n 1024 --no-lpt --steps 3 --warmup 1
run 1024_fp32 --n 1024 --field-bytes 4 --steps 3 --warmup 1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/cfg_*.json")):
    d = json.load(open(f))
    print("%-12s %8.2f ms/step  %.3e cells/s  design bytes / 8 TB/s %.2f  %5.1f GB" % (f.split("cfg_")[1][:-5], d["ms_per_step"], d["value"],
          d["path_roofline"]["frac_of_hbm_peak_design"], d["config"]["device_GB"]))
PY
