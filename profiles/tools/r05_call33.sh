#!/bin/bash
( time timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "grid_200" 2>&1 | tail -5 ) 2>&1 | tail -9
