#!/bin/bash
timeout 1200 python3 -m pytest tests/test_gpu_lines.py -x -q 2>&1 | tail -2
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "general or slab" 2>&1 | tail -2
for n in 720 360; do
AB_ARGS="--n $n" AB_STEPS=3 bash profiles/tools/ab.sh default nokeeprt 2>&1 | grep "ms per step"
done
