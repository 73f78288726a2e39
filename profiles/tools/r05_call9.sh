#!/bin/bash
mkdir -p gpurun_out/r05
AB_ARGS="--exact-steps 0 --table-steps 2" AB_STEPS=3 bash profiles/tools/ab.sh default q1024 q256 noq > gpurun_out/r05/ab_solve3.txt 2>&1
cat gpurun_out/r05/ab_solve3.txt
timeout 600 python3 profiles/tools/fmax_contract.py 128 > gpurun_out/r05/contract128q.txt 2>&1; cat gpurun_out/r05/contract128q.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "full_path or edge_case or solve_beside or invariant" > gpurun_out/r05/parity_q.log 2>&1
tail -8 gpurun_out/r05/parity_q.log
bash profiles/tools/r05_call8.sh
