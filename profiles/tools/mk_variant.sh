#!/bin/bash
# One-object A/B build: mk_variant.sh <name> <file.hip> "<extra flags>"  ->  pinocchio_amd/csrc/build_<name>/libpinfmax_hip_<name>.so
# (the default library's other objects are reused; run `make -C pinocchio_amd/csrc` first).  git-ignored scratch, loaded through PINFMAX_LIB.
set -e
cd "$(dirname "$0")/../../pinocchio_amd/csrc"
name=$1; src=$2; extra=$3
mkdir -p build_$name
base=$(basename $src .hip)
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value"
if [ "$base" = pf_cell_kernels ]; then flags="$flags -ffp-contract=off"; else flags="$flags -ffp-contract=on -DPF_FP_CONTRACT_ON"; fi
/opt/rocm/bin/hipcc $flags $extra -c $src -o build_$name/$base.o
objs=""
for o in pf_api pf_fft_kernels pf_fft16_kernels pf_mixed_kernels pf_cell_kernels pf_synth pf_genic pf_select_sort pf_fabric pf_gfft pf_rccl; do
  if [ "$o" = "$base" ]; then objs="$objs build_$name/$o.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_$name/libpinfmax_hip_$name.so $objs -ldl
echo built build_$name/libpinfmax_hip_$name.so
