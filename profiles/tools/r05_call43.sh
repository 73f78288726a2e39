#!/bin/bash
mkdir -p gpurun_out/r05
for z in 0 1; do
for rep in 1 0; do
PF_SOLVE_BESIDE_Z=$z PF_REPLICATE_DK=$rep python3 bench.py --n 2048 --slab-of 8 --field-bytes 4 --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 --table-steps 0 > gpurun_out/r05_slab_2048_p8_fp32_rep${rep}_beside${z}.json 2> gpurun_out/r05/rep_err.txt || tail -3 gpurun_out/r05/rep_err.txt
done; done
ls -la gpurun_out/r05_slab_2048_p8_fp32_rep*
