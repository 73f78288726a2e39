#!/bin/bash
# round 6, call 1: what the box is (cores, memory), the hand-off probe, and where the -m gpu suite spends its time
mkdir -p gpurun_out/r06
{ nproc; free -g; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null; ulimit -l; } > gpurun_out/r06/box.txt 2>&1
timeout 600 profiles/tools/bin/handoff_probe 16 > gpurun_out/r06/handoff_probe.jsonl 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=60 > gpurun_out/r06/gpu_suite_durations.txt 2>&1
tail -5 gpurun_out/r06/gpu_suite_durations.txt
cat gpurun_out/r06/box.txt gpurun_out/r06/handoff_probe.jsonl
