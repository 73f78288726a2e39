#!/bin/bash
# round 6, call 5: the mixed-radix sizes -- line tests of the new compiled-in plans and of the invariant z-pass with reducing waves, then
# A/B on one box: reducing waves off / on / two extra waves at 768^3, 200^3, 720^3, 1000^3; then the two big-box tests on the faster plane oracle
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_lines.py -x -q > gpurun_out/r06/mixed_lines.txt 2>&1; tail -3 gpurun_out/r06/mixed_lines.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "200 or 96 or 24 or 40 or mixed or grid" > gpurun_out/r06/mixed_parity.txt 2>&1; tail -3 gpurun_out/r06/mixed_parity.txt
for n in 768 200 720 1000; do
  AB_STEPS=3 AB_ARGS="--n $n --exact-steps 0 --boundary 0" bash profiles/tools/ab.sh nospec default extra2 > gpurun_out/r06/ab_mixed_$n.txt 2>&1
  grep -E "zpass_c2r_hess_6to3inv|ms per step|ms per launch" gpurun_out/r06/ab_mixed_$n.txt | sed "s/^/$n: /"
done
AB_STEPS=3 AB_ARGS="--n 768 --field-bytes 4 --exact-steps 0 --boundary 0" bash profiles/tools/ab.sh nospec default > gpurun_out/r06/ab_mixed_768_fp32.txt 2>&1; grep -E "zpass|ms per step" gpurun_out/r06/ab_mixed_768_fp32.txt | sed "s/^/768 fp32: /"
timeout 900 python3 -m pytest tests/test_gpu_config5.py tests/test_lpt_analytic.py -x -q -m gpu --durations=4 -k "config5 or full_bench" > gpurun_out/r06/bigbox_tests2.txt 2>&1; tail -8 gpurun_out/r06/bigbox_tests2.txt
