#!/bin/bash
mkdir -p gpurun_out/r05
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "general or example_size_200 or grid_200 or release_their_memory" > gpurun_out/r05/general.log 2>&1
tail -15 gpurun_out/r05/general.log
timeout 900 python3 -m pytest tests/test_gpu_gloo_ranks.py -x -q > gpurun_out/r05/gloo.log 2>&1
tail -15 gpurun_out/r05/gloo.log
