#!/bin/bash
# Round artefacts in one gpurun call (run from the repo root ON THE GPU BOX):
#   profiles/tools/collect.sh r03                                  (the metric's configuration)
#   BENCH_ARGS="--field-bytes 4" PF_SUMMARY_FB=4 profiles/tools/collect.sh r03_fp32   (another one: extra bench arguments, and what summarise.py stamps)
# 1. smoke()  2. rocprofv3 --kernel-trace --stats of a bench command, as the bench runs it and once more with every kernel in line
# 3. counter passes of one bench
# step, each in its own run with --kernel-trace only (gpurun refuses --pmc together with other trace domains):
# FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_WAVES | SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES | GRBM_GUI_ACTIVE
# 4. profiles/tools/summarise.py -> gpurun_out/<round>_*.json / .csv, to be copied into profiles/ and committed
# 5. the default bench line (python3 bench.py), which then carries the counters of step 3.
# The profiled program is `python3 bench.py ...` directly after `--` (no env / bash -c hop: the profiler has initialised the GPU).
tag=${1:-r02}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
rm -rf gpurun_out/prof_${tag} gpurun_out/pmc_${tag}_*
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-n 0 --exact-steps 0 --table-steps 0 $BENCH_ARGS \
  > $R/gpurun_out/${tag}_bench_profiled.json 2> $R/gpurun_out/${tag}_bench_profiled.err
# the same command with every kernel in line (PF_SOLVE_BESIDE_Z=0, exported by this shell and inherited by the profiled program):
# in the default order the solve of sweep radius i runs beside the z-pass of radius i + 1 and the durations of those two kernels
# in the trace above overlap; here every duration is the kernel's own.  The counter passes below run in line too (the same
# kernels doing the same work; a counter pass serialises the dispatches anyway).
export PF_SOLVE_BESIDE_Z=0
rm -rf $R/gpurun_out/prof_${tag}_inline
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_inline -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-n 0 --exact-steps 0 $BENCH_ARGS \
  > $R/gpurun_out/${tag}_bench_profiled_inline.json 2> $R/gpurun_out/${tag}_bench_profiled_inline.err
pmc() {  # pmc <name> <counters...>
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-n 0 --exact-steps 0 $BENCH_ARGS \
    > $R/gpurun_out/pmc_${tag}_$name.out 2> $R/gpurun_out/pmc_${tag}_$name.err
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc valu SQ_INSTS_VALU SQ_WAVES
pmc busy SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
pmc clk GRBM_GUI_ACTIVE
[ -n "$PF_COLLECT_LDS" ] && pmc lds SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
unset PF_SOLVE_BESIDE_Z
cd $R
python3 profiles/tools/summarise.py $tag
# the default bench line last, with the counter summaries of THIS box and THESE kernel sources in place (bench.py attaches
# roofline.traffic / valu only from profiles stamped with the sources it runs)
cp gpurun_out/${tag}_pmc_traffic.json gpurun_out/${tag}_pmc_valu.json gpurun_out/${tag}_kernel_stats.csv gpurun_out/${tag}_kernel_stats_inline.csv profiles/
[ -n "$PROFILE_ROUND" ] || export PROFILE_ROUND=${tag%_fp32}
timeout 1500 python3 bench.py $BENCH_ARGS > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
tail -c 400 gpurun_out/${tag}_bench_default.json; echo
