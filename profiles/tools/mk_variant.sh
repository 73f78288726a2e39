#!/bin/bash
# One-source A/B build: mk_variant.sh <name> <file.hip> "<extra flags>"  ->  pinocchio_amd/csrc/build_<name>/libpinfmax_hip_<name>.so
# (the default library's other objects are reused; run `make -C pinocchio_amd/csrc` first).  git-ignored scratch, loaded through PINFMAX_LIB.
# pf_mixed_kernels.hip is three objects (PF_MIXED_PART 0, 1, 2): all three are rebuilt with the flags.
set -e
cd "$(dirname "$0")/../../pinocchio_amd/csrc"
name=$1; src=$2; extra=$3
mkdir -p build_$name
base=$(basename $src .hip)
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value"
if [ "$base" = pf_cell_kernels ]; then flags="$flags -ffp-contract=off"; else flags="$flags -ffp-contract=on -DPF_FP_CONTRACT_ON"; fi
if [ "$base" = pf_mixed_kernels ]; then
  /opt/rocm/bin/hipcc $flags $extra -c $src -o build_$name/$base.o &
  /opt/rocm/bin/hipcc $flags $extra -DPF_MIXED_PART=1 -c $src -o build_$name/${base}_p1.o &
  /opt/rocm/bin/hipcc $flags $extra -DPF_MIXED_PART=2 -c $src -o build_$name/${base}_p2.o &
  wait
else
  /opt/rocm/bin/hipcc $flags $extra -c $src -o build_$name/$base.o
fi
objs=""
for o in pf_api pf_fft_kernels pf_fft16_kernels pf_mixed_kernels pf_mixed_kernels_p1 pf_mixed_kernels_p2 pf_cell_kernels pf_synth pf_genic pf_select_sort pf_fabric pf_gfft pf_rccl; do
  if [ -f build_$name/$o.o ]; then objs="$objs build_$name/$o.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_$name/libpinfmax_hip_$name.so $objs -ldl
echo built build_$name/libpinfmax_hip_$name.so
