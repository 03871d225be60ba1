#!/bin/bash
# A/B of the fast kernel's head dot products: two v_fma_f32 (shipped since round 2) against one v_pk_fma_f32 (round 1), FULL form
set -e
cd /root/repo
B=ibl-nerf_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -mllvm -amdgpu-mfma-vgpr-form -DIBL_MX_VARIANT=0 -DIBL_MX_HEADS_PK_FMA -c ibl-nerf_amd/csrc/mlp_kernel_mx.hip -o scratch/mx_full_pkfma.o
objs=$(ls $B/*.o | grep -v mlp_kernel_mx_full.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_mx_pkfma.so scratch/mx_full_pkfma.o $objs
ls -la scratch/lib_mx_pkfma.so
