#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_lines.py -x -q -k "several_jobs or 768 or 200 or 384 or 120" 2>&1 | tail -4
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "general or example_size_200 or grid_200" 2>&1 | tail -3
AB_ARGS="--n 768" AB_STEPS=2 bash profiles/tools/ab.sh default noct nokeep 2>&1 | tail -20
AB_ARGS="--n 200" AB_STEPS=5 bash profiles/tools/ab.sh default noct 2>&1 | tail -20
AB_ARGS="--n 768 --field-bytes 4" AB_STEPS=2 bash profiles/tools/ab.sh default noct 2>&1 | tail -20
