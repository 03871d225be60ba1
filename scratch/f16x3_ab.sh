#!/bin/bash
# A/B builds of the f16x3 TRUNK kernel into scratch/lib_f16x3_<tag>.so (timing experiments; the other objects come from the product build)
set -e
cd /root/repo
B=ibl-nerf_amd/build
build() {  # tag, extra flags...
  tag=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DIBL_F16X3 -DIBL_VARIANT=1 "$@" -c ibl-nerf_amd/csrc/mlp_kernel.hip -o scratch/f16x3_trunk_$tag.o
  objs=$(ls $B/*.o | grep -v mlp_kernel_f16x3_trunk.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_f16x3_$tag.so scratch/f16x3_trunk_$tag.o $objs
}
build relubits -DIBL_ABLATE_RELU_BITS &
build vgprform -mllvm -amdgpu-mfma-vgpr-form &
wait
ls -la scratch/lib_f16x3_*.so
