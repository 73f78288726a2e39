#!/bin/bash
BENCH_ARGS="--n 768" PF_SUMMARY_N=768 PF_COLLECT_LDS=1 bash profiles/tools/collect.sh r05_768 2>&1 | tail -5
