#!/bin/bash
mkdir -p gpurun_out/r05
PINFMAX_LIB=$PWD/pinocchio_amd/csrc/build_twlds/libpinfmax_hip_twlds.so timeout 300 python3 -m pytest tests/test_gpu_lines.py -x -q -k "strided_pass_lines and 1024-8 or first_pass_filter and 1024-8" 2>&1 | tail -3
AB_ARGS="--exact-steps 0 --table-steps 2" AB_STEPS=3 bash profiles/tools/ab.sh default twlds default twlds > gpurun_out/r05/ab_twlds.txt 2>&1
cat gpurun_out/r05/ab_twlds.txt
timeout 600 python3 -m pytest tests/test_gpu_multirank.py -x -q -k "slab_ranks_match_single_rank" 2>&1 | tail -3
