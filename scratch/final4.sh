#!/bin/bash
# final artefacts of the round: smoke, default bench line, rocprofv3 kernel stats of the bench command, counters
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 300 gpurun_out/final_bench.json
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/prof_final5 gpurun_out/pmc_valu gpurun_out/pmc_busy gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_clk
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final5 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-n 0 > $R/gpurun_out/prof_final5.out 2> $R/gpurun_out/prof_final5.err
cd $R
ls gpurun_out/prof_final5/*/ | head -5
export PF_PRUNE_EPS=0
bash scratch/pmc.sh valu SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU
bash scratch/pmc.sh busy SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
bash scratch/pmc.sh fetch FETCH_SIZE
bash scratch/pmc.sh write WRITE_SIZE
python scratch/pmc_summ.py valu busy
python scratch/pmc_traffic.py
bash scratch/clk.sh
