#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_lines.py -x -q -k "24 or 40 or 96 or 120 or 200 or 384 or 768 or 1000 or 1536 or 2000" 2>&1 | tail -3
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "invariant_zpass or general or example_size_200 or grid_200" 2>&1 | tail -3
AB_ARGS="--n 768" AB_STEPS=2 bash profiles/tools/ab.sh default twtab nopad 2>&1 | tail -16
AB_ARGS="--n 200" AB_STEPS=5 bash profiles/tools/ab.sh default twtab nopad 2>&1 | tail -16
AB_ARGS="--n 768 --field-bytes 4" AB_STEPS=2 bash profiles/tools/ab.sh default twtab nopad 2>&1 | tail -16
