#!/bin/bash
mkdir -p gpurun_out/r05
timeout 300 ./profiles/tools/bin/valu_probe > gpurun_out/r05/valu_probe.jsonl 2> gpurun_out/r05/valu_probe.err
cat gpurun_out/r05/valu_probe.jsonl
