#!/bin/bash
# A/B of kernel variants on ONE box (run from the repo root on the GPU box): ab_kernels.sh <probe.py> name1 name2 ...
# every name is a library built by mk_variant.sh ("default" = the product library); the probe runs once per library in its own process
cd "$(dirname "$0")/../.." || exit 1
probe=$1; shift
for v in "$@"; do
  lib=$PWD/pinocchio_amd/csrc/build_$v/libpinfmax_hip_$v.so
  [ "$v" = default ] && lib=$PWD/pinocchio_amd/libpinfmax_hip.so
  echo "== $v"
  PINFMAX_LIB=$lib python3 $probe 2>&1 | tail -${AB_TAIL:-6}
done
