#!/bin/bash
timeout 300 python scratch/cmicro.py
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "collapse_cells or full_path or golden or hmf or scale_dependent" 2>&1 | tail -3
python - <<'PY'
import time, numpy as np
from pinocchio_amd import api, synth
x, y = synth.invgrow_table("lcdm")
with api.Fmax(64) as f:
    f.set_invgrow(x, y)
    f.set_collapse_model(1, [0.25, 0.75, 0, 0], [1.28e-5])
    f.ct_build(0, 1.7); t0 = time.perf_counter(); f.ct_build(0, 1.7); t1 = time.perf_counter()
    print("ELL_SNG table of one radius: %.3f s" % (t1 - t0))
    f.set_collapse_model(0)
    f.ct_build(0, 1.7); t0 = time.perf_counter(); f.ct_build(0, 1.7); t1 = time.perf_counter()
    print("ELL_CLASSIC table of one radius: %.4f s" % (t1 - t0))
PY
