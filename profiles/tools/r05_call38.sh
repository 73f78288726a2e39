#!/bin/bash
timeout 1200 python3 -m pytest tests/test_gpu_lines.py -x -q -k "several_jobs or 768 or 200 or 640 or 1000" 2>&1 | tail -3
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "invariant_zpass or general or example_size_200 or grid_200" 2>&1 | tail -3
for n in 768 200 640; do
st=2; [ $n = 200 ] && st=5
AB_ARGS="--n $n" AB_STEPS=$st bash profiles/tools/ab.sh default nonext default nonext 2>&1 | grep "ms per\|[xy]pass"
done
AB_ARGS="--n 768 --field-bytes 4" AB_STEPS=2 bash profiles/tools/ab.sh default nonext 2>&1 | grep "ms per\|[xy]pass"
