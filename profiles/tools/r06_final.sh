#!/bin/bash
# round 6, the committed artefacts in one call (run from the repo root ON THE GPU BOX, after the last change to pinocchio_amd/csrc):
#   goldens of the bench configurations on these sources; collect.sh for the metric's configuration, for fp32 fields on one GPU, for
#   BASELINE config 5's slab and for 768^3; the slab matrix; the other sizes; the config-5 test's report.  Everything lands in gpurun_out/
#   (merged back by gpurun), to be copied into profiles/.
mkdir -p gpurun_out/r06
python3 tests/golden/make_bench_fingerprints.py > gpurun_out/r06/goldens.log 2>&1; tail -3 gpurun_out/r06/goldens.log
cp tests/golden/bench_fingerprints.json gpurun_out/r06/bench_fingerprints.json
bash profiles/tools/collect.sh r06 > gpurun_out/r06/collect_r06.log 2>&1; tail -2 gpurun_out/r06/collect_r06.log
BENCH_ARGS="--field-bytes 4" PF_SUMMARY_FB=4 bash profiles/tools/collect.sh r06_fp32 > gpurun_out/r06/collect_r06_fp32.log 2>&1; tail -2 gpurun_out/r06/collect_r06_fp32.log
BENCH_ARGS="--slab-of 8 --n 2048 --field-bytes 4" PF_SUMMARY_N=2048 PF_SUMMARY_FB=4 PF_SUMMARY_SLAB_OF=8 PROFILE_ROUND=r06 bash profiles/tools/collect.sh r06_2048 > gpurun_out/r06/collect_r06_2048.log 2>&1; tail -2 gpurun_out/r06/collect_r06_2048.log
bash profiles/tools/slab_matrix.sh r06 > gpurun_out/r06/slab_matrix.log 2>&1; tail -14 gpurun_out/r06/slab_matrix.log
PF_SOLVE_BESIDE_Z=0 timeout 600 python3 bench.py --slab-of 8 --n 2048 --field-bytes 4 --steps 3 --warmup 1 > gpurun_out/r06_slab_2048_p8_fp32_inline.json 2> gpurun_out/r06/slab2048_inline.err
PF_REPLICATE_DK=1 PF_SOLVE_BESIDE_Z=0 timeout 600 python3 bench.py --slab-of 8 --n 2048 --field-bytes 4 --steps 3 --warmup 1 > gpurun_out/r06_slab_2048_p8_fp32_physical_inline.json 2> gpurun_out/r06/slab2048_phys.err
BENCH_ARGS="--n 768" PF_SUMMARY_N=768 bash profiles/tools/collect.sh r06_768 > gpurun_out/r06/collect_r06_768.log 2>&1; tail -2 gpurun_out/r06/collect_r06_768.log
timeout 900 python3 bench.py --n 768 --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 > gpurun_out/r06_bench_768.json 2> gpurun_out/r06/bench_768.err
timeout 900 python3 bench.py --n 768 --field-bytes 4 --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 > gpurun_out/r06_bench_768_fp32.json 2> gpurun_out/r06/bench_768_fp32.err
timeout 900 python3 bench.py --n 1000 --steps 2 --warmup 1 --cpu-n 0 --exact-steps 0 > gpurun_out/r06_bench_1000.json 2> gpurun_out/r06/bench_1000.err
timeout 300 python3 bench.py --n 640 --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 > gpurun_out/r06_bench_640.json 2> gpurun_out/r06/bench_640.err
timeout 300 python3 bench.py --n 720 --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 > gpurun_out/r06_bench_720.json 2> gpurun_out/r06/bench_720.err
timeout 300 python3 bench.py --n 512 --steps 5 --warmup 1 --cpu-n 0 --exact-steps 0 > gpurun_out/r06_bench_512.json 2> gpurun_out/r06/bench_512.err
timeout 300 python3 bench.py --n 256 --steps 5 --warmup 1 --cpu-n 0 --exact-steps 0 > gpurun_out/r06_bench_256.json 2> gpurun_out/r06/bench_256.err
timeout 300 python3 bench.py --n 200 --steps 5 --warmup 1 --cpu-n 0 --exact-steps 0 > gpurun_out/r06_bench_200.json 2> gpurun_out/r06/bench_200.err
PF_GENERAL=1 timeout 300 python3 bench.py --n 200 --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 > gpurun_out/r06_bench_200_chirpz.json 2> gpurun_out/r06/bench_200_chirpz.err
timeout 600 python3 -m pytest tests/test_gpu_config5.py -x -q -s > gpurun_out/r06/config5_final.txt 2>&1; tail -2 gpurun_out/r06/config5_final.txt
ls gpurun_out | head -80
