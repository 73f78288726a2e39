#!/bin/bash
# round 6, call 2: the new tests (config 5 on its own workload, the loopback slab's known answers, the chirp-z line tap), the hand-off
# paths through the existing product tests, and the bench line with its `boundary` object
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_config5.py -x -q -s --durations=5 > gpurun_out/r06/config5_test.txt 2>&1; tail -5 gpurun_out/r06/config5_test.txt
timeout 900 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_lines.py -x -q -k "loopback or chirp or rccl or torch_exchange" --durations=8 > gpurun_out/r06/new_tests.txt 2>&1; tail -12 gpurun_out/r06/new_tests.txt
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_compat_host.py tests/test_examples.py -x -q -m gpu --durations=8 > gpurun_out/r06/handoff_tests.txt 2>&1; tail -12 gpurun_out/r06/handoff_tests.txt
timeout 900 python3 bench.py --steps 5 --warmup 1 --cpu-n 0 --exact-steps 0 > gpurun_out/r06/bench_boundary.json 2> gpurun_out/r06/bench_boundary.err; tail -3 gpurun_out/r06/bench_boundary.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_boundary.json"))
print("ms_per_step", d["ms_per_step"], "boundary", json.dumps(d.get("boundary")))
PY
PF_HOST_REGISTER=1 timeout 900 python3 bench.py --steps 2 --warmup 1 --cpu-n 0 --exact-steps 0 --table-steps 0 > gpurun_out/r06/bench_boundary_registered.json 2> gpurun_out/r06/bench_boundary_registered.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_boundary_registered.json"))
print("registered: ms_per_step", d["ms_per_step"], "boundary", json.dumps(d.get("boundary")))
PY
