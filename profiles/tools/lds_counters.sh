#!/bin/bash
# LDS / wait-side counters of one bench step (run from the repo root ON THE GPU BOX):  profiles/tools/lds_counters.sh r03
# Each group is its own rocprofv3 run with --kernel-trace only (gpurun refuses --pmc together with other trace domains);
# the profiled program is `python3 bench.py ...` directly after `--`.  Output: gpurun_out/pmc_<tag>_lds*/ and, through
# profiles/tools/summarise_lds.py, gpurun_out/<tag>_pmc_lds.json (copy into profiles/ to commit).
tag=${1:-r03}
R=$PWD
export TMPDIR=/tmp
export PF_SOLVE_BESIDE_Z=0  # every kernel in line, as in the counter passes of collect.sh (the same kernels doing the same work)
mkdir -p gpurun_out
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $R/gpurun_out/${tag}_sq_counter_names.txt
pmc() {  # pmc <name> <counters...>
  name=$1; shift
  rm -rf $R/gpurun_out/pmc_${tag}_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-n 0 --exact-steps 0 $BENCH_ARGS \
    > $R/gpurun_out/pmc_${tag}_$name.out 2> $R/gpurun_out/pmc_${tag}_$name.err || tail -3 $R/gpurun_out/pmc_${tag}_$name.err
}
pmc lds1 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pmc lds2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc lds3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU
pmc lds4 SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
cd $R
python3 profiles/tools/summarise_lds.py $tag
