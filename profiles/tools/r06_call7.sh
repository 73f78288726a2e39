#!/bin/bash
# round 6, call 7: the two-rows-per-thread invariant z-pass of fp32 fields (k_c2r_invariants_pk2): line tests and fp32 parity first, then
# A/B against the one-row kernel (build nopk2) on one box: fp32 1024^3, BASELINE config 5's slab (every kernel in line, and the default order), fp32 512^3
mkdir -p gpurun_out/r06
timeout 600 python3 -m pytest tests/test_gpu_lines.py -x -q -k "invariant" > gpurun_out/r06/pk2_lines.txt 2>&1; tail -3 gpurun_out/r06/pk2_lines.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_config5.py -x -q -k "fp32 or f32 or small_box or invariant" > gpurun_out/r06/pk2_parity.txt 2>&1; tail -3 gpurun_out/r06/pk2_parity.txt
AB_STEPS=4 AB_ARGS="--field-bytes 4 --exact-steps 0 --boundary 0" bash profiles/tools/ab.sh nopk2 default > gpurun_out/r06/ab_pk2_1024.txt 2>&1; grep -E "zpass|collapse_inv|ms per step|ms per launch" gpurun_out/r06/ab_pk2_1024.txt | sed "s/^/1024 fp32: /"
AB_STEPS=4 AB_ARGS="--n 512 --field-bytes 4 --exact-steps 0 --boundary 0" bash profiles/tools/ab.sh nopk2 default > gpurun_out/r06/ab_pk2_512.txt 2>&1; grep -E "zpass_c2r_hess_6to3inv|ms per step" gpurun_out/r06/ab_pk2_512.txt | sed "s/^/512 fp32: /"
for v in nopk2 default; do
  lib=pinocchio_amd/csrc/build_$v/libpinfmax_hip_$v.so; [ "$v" = default ] && lib=pinocchio_amd/libpinfmax_hip.so
  for inl in 0 1; do
    envs="PINFMAX_LIB=$PWD/$lib"; [ $inl = 1 ] && envs="$envs PF_SOLVE_BESIDE_Z=0"
    env $envs timeout 600 python3 bench.py --slab-of 8 --n 2048 --field-bytes 4 --steps 3 --warmup 1 > gpurun_out/r06/pk2_slab_${v}_$inl.json 2> gpurun_out/r06/pk2_slab_${v}_$inl.err
  done
done
python3 - <<'PY'
import json
for v in ("nopk2", "default"):
    for inl in (0, 1):
        try:
            d = json.load(open(f"gpurun_out/r06/pk2_slab_{v}_{inl}.json"))
        except Exception as e:
            print(v, inl, "failed", e); continue
        z = [k for k in d["kernels"] if k["name"] == "zpass_c2r_hess_6to3inv"][0]
        print("slab 2048", v, "inline" if inl else "default order", "ms_per_step %.1f" % d["ms_per_step"], "z-pass %.1f ms/step %.0f GB/s" % (z["ms_per_step"], z["GBps"]))
PY
