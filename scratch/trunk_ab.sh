#!/bin/bash
# A/B build: the two TRUNK kernels of the default mode without the point-generation branch in their input stage (-DIBL_NO_POINT_GEN),
# linked with the product's other objects into scratch/lib_nogen.so.  Run from the repo root, then on the GPU: python scratch/trunk_ab.py
set -e
B=ibl-nerf_amd/build; C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude"
/opt/rocm/bin/hipcc $C -mllvm -amdgpu-mfma-vgpr-form -DIBL_MX_VARIANT=5 -DIBL_NO_POINT_GEN -c ibl-nerf_amd/csrc/mlp_kernel_mx.hip -o scratch/nogen_mx_trunk_x.o &
/opt/rocm/bin/hipcc $C -DIBL_F16X3 -DIBL_VARIANT=1 -DIBL_NO_POINT_GEN -c ibl-nerf_amd/csrc/mlp_kernel.hip -o scratch/nogen_f16x3_trunk.o &
wait
OBJS=$(ls $B/*.o | grep -v "mlp_kernel_mx_trunk_x.o\|mlp_kernel_f16x3_trunk.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_nogen.so $OBJS scratch/nogen_mx_trunk_x.o scratch/nogen_f16x3_trunk.o
ls -la scratch/lib_nogen.so
