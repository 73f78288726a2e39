#!/bin/bash
# round 6, call 3: config 5 on its own workload (small box first), then the fp32-field contract from data
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_config5.py -x -q -s --durations=5 > gpurun_out/r06/config5_test.txt 2>&1; tail -5 gpurun_out/r06/config5_test.txt
timeout 1500 python3 profiles/tools/fp32_contract.py > gpurun_out/r06/fp32_contract.json 2> gpurun_out/r06/fp32_contract.err; tail -3 gpurun_out/r06/fp32_contract.err
