#!/bin/bash
timeout 1200 python3 -m pytest tests/test_gpu_lines.py -x -q -k invariant 2>&1 | tail -3
for n in 768 200; do
st=2; [ $n = 200 ] && st=5
AB_ARGS="--n $n" AB_STEPS=$st bash profiles/tools/ab.sh default walk 2>&1 | tail -15
done
export PF_INVARIANTS=0 PF_LPT_FUSE=0
for n in 768 200; do
st=2; [ $n = 200 ] && st=5
AB_ARGS="--n $n" AB_STEPS=$st bash profiles/tools/ab.sh default 2>&1 | tail -15
done
