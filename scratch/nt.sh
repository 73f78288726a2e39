#!/bin/bash
# non-temporal access experiment: default build, then the FFT kernels rebuilt with -DPF_NT on this box
for i in 1 2; do timeout 300 python scratch/zmicro.py; done
cd pinocchio_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-function -Wno-unused-value -DPF_NT -c pf_fft_kernels.hip -o pf_fft_kernels.o
make 2>&1 | tail -1
cd ../..
echo "== PF_NT"
for i in 1 2; do timeout 300 python scratch/zmicro.py; done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "transforms or second_derivatives" 2>&1 | tail -2
