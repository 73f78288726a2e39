#!/bin/bash
mkdir -p gpurun_out/r05
( time timeout 2400 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -4 ) > gpurun_out/r05/full_gpu_suite.log 2>&1
tail -8 gpurun_out/r05/full_gpu_suite.log
