#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "collapse_cells or full_path or golden or hmf" 2>&1 | tail -3
for v in 3 6 8 12; do PF_COLLAPSE_WG_PER_CU=$v timeout 300 python scratch/cmicro.py; done
