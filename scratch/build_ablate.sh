#!/bin/bash
# builds ablation variants of the library into scratch/ (timing experiments only)
set -e
cd /root/repo
B=ibl-nerf_amd/build
for v in ${ABL:-NO_LOADS NO_MFMA NO_BARRIER NO_FRAG NO_EPI}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DIBL_ABLATE_$v -c ibl-nerf_amd/csrc/mlp_kernel.hip -o scratch/mlp_$v.o &
done
wait
for v in ${ABL:-NO_LOADS NO_MFMA NO_BARRIER NO_FRAG NO_EPI}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_$v.so scratch/mlp_$v.o $B/mlp_kernel_mx_full.o $B/mlp_kernel_mx_trunk.o $B/mlp_kernel_mx_refl.o $B/mlp_kernel_mx_full_ci.o $B/mlp_kernel_mx_refl_ci.o $B/render_kernels.hip.o $B/pack_kernels.hip.o $B/api.cpp.o $B/pack.cpp.o
done
ls -la scratch/*.so
