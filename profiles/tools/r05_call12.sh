#!/bin/bash
mkdir -p gpurun_out/r05
timeout 300 python3 -m pytest tests/test_gpu_lines.py -x -q -k "chirp" > gpurun_out/r05/chirp.log 2>&1; tail -4 gpurun_out/r05/chirp.log
( time timeout 2400 python3 -m pytest tests -m gpu -q --durations=25 ) > gpurun_out/r05/full_gpu.log 2>&1
tail -60 gpurun_out/r05/full_gpu.log
