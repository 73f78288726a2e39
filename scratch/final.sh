#!/bin/bash
# final round check: whole GPU suite, smoke, default bench line, rocprofv3 kernel stats of the same command
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/final_tests.log 2>&1; tail -3 gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 400 gpurun_out/final_bench.json
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final2 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-n 0 > $R/gpurun_out/prof_final2.out 2> $R/gpurun_out/prof_final2.err
cd $R
ls gpurun_out/prof_final2/*/ | head -5
