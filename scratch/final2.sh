#!/bin/bash
# final artefacts of the round: smoke, default bench line, rocprofv3 kernel stats of the bench command
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 300 gpurun_out/final_bench.json
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final3 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-n 0 > $R/gpurun_out/prof_final3.out 2> $R/gpurun_out/prof_final3.err
cd $R
ls gpurun_out/prof_final3/*/ | head -5
