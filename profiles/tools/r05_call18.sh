#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_lines.py -x -q -k "invariant" 2>&1 | tail -3
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "invariant_zpass" 2>&1 | tail -3
AB_ARGS="--n 768" AB_STEPS=2 bash profiles/tools/ab.sh default nopre 2>&1 | tail -16
AB_ARGS="--n 200" AB_STEPS=5 bash profiles/tools/ab.sh default nopre 2>&1 | tail -16
