#!/bin/bash
# round-1 final profiles: kernel trace + stats of the default bench command, then HBM traffic counters (separate passes)
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-n 0 > $R/gpurun_out/prof_final.out 2> $R/gpurun_out/prof_final.err
cd $R
bash scratch/pmc.sh fetch FETCH_SIZE
bash scratch/pmc.sh write WRITE_SIZE
ls gpurun_out/prof_final/*/ | head
python scratch/pcie.py
