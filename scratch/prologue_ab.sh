#!/bin/bash
# Timing ablation: how much of the mixed TRUNK kernel is its per-group input stage (point load / generation + encoding, serial before the first MFMA)?
set -e
B=ibl-nerf_amd/build; C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Iinclude -mllvm -amdgpu-mfma-vgpr-form"
OBJS=$(ls $B/*.o | grep -v "mlp_kernel_mx_trunk_x.o\|mlp_kernel_mx_trunk.o")
/opt/rocm/bin/hipcc $C -DIBL_MX_VARIANT=5 -DIBL_MX_ABLATE_PROLOGUE -c ibl-nerf_amd/csrc/mlp_kernel_mx.hip -o scratch/nopro_x.o &
/opt/rocm/bin/hipcc $C -DIBL_MX_VARIANT=1 -DIBL_MX_ABLATE_PROLOGUE -c ibl-nerf_amd/csrc/mlp_kernel_mx.hip -o scratch/nopro_t.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_noprologue.so $OBJS scratch/nopro_x.o scratch/nopro_t.o
ls -la scratch/lib_noprologue.so
