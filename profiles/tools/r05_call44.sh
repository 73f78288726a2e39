#!/bin/bash
timeout 1200 python3 -m pytest tests/test_gpu_lines.py -x -q 2>&1 | tail -2
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "invariant_zpass or general or example_size_200 or grid_200 or slab_ranks" 2>&1 | tail -2
for n in 720 360 1200; do
AB_ARGS="--n $n" AB_STEPS=2 bash profiles/tools/ab.sh default nokeeprt 2>&1 | grep "ms per\|[xy]pass"
done
AB_ARGS="--n 720 --field-bytes 4" AB_STEPS=2 bash profiles/tools/ab.sh default nokeeprt 2>&1 | grep "ms per\|[xy]pass"
