#!/bin/bash
# ab_solve_stream.sh <values of PF_SOLVE_BESIDE_Z ...>: the bench step with the sweep's solve beside the next z-pass (1) or in line (0), one process per setting
cd "$(dirname "$0")/../.." || exit 1
for v in "$@"; do
  PF_SOLVE_BESIDE_Z=$v python3 profiles/tools/solve_stream_probe.py 2>&1 | tail -${AB_TAIL:-1}
done
