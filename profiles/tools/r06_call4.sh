#!/bin/bash
# round 6, call 4: the packed fp32 algebra of every pfc<float> transform -- correctness first (line tests, fp32 parity, fp32 slabs), then
# what it buys (fp32 1024^3, config 5's slab, 768^3 fp32), then the two big-box tests with the oracle on the cores the cgroup grants
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_lines.py -x -q > gpurun_out/r06/pk_lines.txt 2>&1; tail -3 gpurun_out/r06/pk_lines.txt
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -k "fp32 or f32 or field_bytes or 4 or loopback" > gpurun_out/r06/pk_parity.txt 2>&1; tail -3 gpurun_out/r06/pk_parity.txt
timeout 600 python3 bench.py --field-bytes 4 --steps 5 --warmup 1 --cpu-n 0 --exact-steps 0 --boundary 0 > gpurun_out/r06/pk_bench_fp32.json 2> gpurun_out/r06/pk_bench_fp32.err
timeout 600 python3 bench.py --slab-of 8 --n 2048 --field-bytes 4 --steps 3 --warmup 1 > gpurun_out/r06/pk_slab_2048.json 2> gpurun_out/r06/pk_slab_2048.err
PF_SOLVE_BESIDE_Z=0 timeout 600 python3 bench.py --slab-of 8 --n 2048 --field-bytes 4 --steps 3 --warmup 1 > gpurun_out/r06/pk_slab_2048_inline.json 2> gpurun_out/r06/pk_slab_2048_inline.err
timeout 600 python3 bench.py --n 768 --field-bytes 4 --steps 3 --warmup 1 --cpu-n 0 --exact-steps 0 --boundary 0 > gpurun_out/r06/pk_bench_768_fp32.json 2> gpurun_out/r06/pk_bench_768_fp32.err
python3 - <<'PY'
import json
for f in ("pk_bench_fp32", "pk_slab_2048", "pk_slab_2048_inline", "pk_bench_768_fp32"):
    try:
        d = json.load(open(f"gpurun_out/r06/{f}.json"))
    except Exception as e:
        print(f, "FAILED", e); continue
    print(f, "ms_per_step", round(d["ms_per_step"], 1), "device_GB", d["config"].get("device_GB"))
    for k in (d.get("kernel_table") or {}).get("kernels", d.get("kernels", [])):
        print("   %-26s %8.2f ms/step %7.0f GB/s" % (k["name"], k["ms_per_step"], k["GBps"]))
PY
timeout 900 python3 -m pytest tests/test_gpu_config5.py tests/test_lpt_analytic.py -x -q -m gpu --durations=6 > gpurun_out/r06/bigbox_tests.txt 2>&1; tail -12 gpurun_out/r06/bigbox_tests.txt
