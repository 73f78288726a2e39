#!/bin/bash
# round 6, call 8: the mixed-radix invariant z-pass with reducing waves once more, on the final sources (the first A/B ran while a change to the
# shared fold had the fp64 kernels of the file at 108 instead of 94 registers -- both sides of it): default (off) / on / on with two extra waves
mkdir -p gpurun_out/r06
for n in 768 720 1000 200; do
  AB_STEPS=3 AB_ARGS="--n $n --exact-steps 0 --boundary 0" bash profiles/tools/ab.sh default spec1 spec2 > gpurun_out/r06/ab2_mixed_$n.txt 2>&1
  grep -E "zpass_c2r_hess_6to3inv|ms per step|ms per launch" gpurun_out/r06/ab2_mixed_$n.txt | sed "s/^/$n: /"
done
