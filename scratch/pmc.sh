#!/bin/bash
# usage: pmc.sh <tag> <counters...>   (run from repo root on the GPU box)
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
cd /tmp
PF_NO_OVERLAP=1 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --n 1024 --ns 2 --no-lpt --steps 1 --warmup 0 --cpu-n 0 > $R/gpurun_out/pmc_$tag.out 2> $R/gpurun_out/pmc_$tag.err
cd $R
