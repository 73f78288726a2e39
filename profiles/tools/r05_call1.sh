#!/bin/bash
# round 5, first call: segment-width probe + the library's config-5 slab and fp32 1024^3 numbers at the start of the round
mkdir -p gpurun_out/r05
timeout 900 ./profiles/tools/bin/seg_probe > gpurun_out/r05/seg_probe.jsonl 2> gpurun_out/r05/seg_probe.err
PF_SOLVE_BESIDE_Z=0 timeout 600 python3 bench.py --slab-of 8 --n 2048 --field-bytes 4 --steps 3 --warmup 1 > gpurun_out/r05/base_slab2048_inline.json 2> gpurun_out/r05/base_slab2048_inline.err
timeout 600 python3 bench.py --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 --field-bytes 4 > gpurun_out/r05/base_fp32_1024.json 2> gpurun_out/r05/base_fp32_1024.err
tail -3 gpurun_out/r05/*.err
cat gpurun_out/r05/seg_probe.jsonl
