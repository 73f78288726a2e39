#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 ./profiles/tools/bin/seg_probe > gpurun_out/r05/seg_probe2.jsonl 2> gpurun_out/r05/seg_probe2.err
tail -2 gpurun_out/r05/seg_probe2.err
