#!/bin/bash
for i in 1 2; do timeout 300 python scratch/zmicro.py; done
cd pinocchio_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-function -Wno-unused-value -DPF_TILE_LDS_KB=128 -c pf_fft_kernels.hip -o pf_fft_kernels.o
make 2>&1 | tail -1 | cut -c1-60
cd ../..
echo "== 128 KB tiles (T=8 at 1024)"
for i in 1 2; do timeout 300 python scratch/zmicro.py; done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "transforms or second_derivatives" 2>&1 | tail -2
