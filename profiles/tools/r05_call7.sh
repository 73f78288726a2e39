#!/bin/bash
mkdir -p gpurun_out/r05
AB_ARGS="--exact-steps 0 --table-steps 2" AB_STEPS=3 bash profiles/tools/ab.sh default w5 b512 b1024 > gpurun_out/r05/ab_solve2.txt 2>&1
cat gpurun_out/r05/ab_solve2.txt
