#!/bin/bash
# A/B of compiler flags on the mixed TRUNK kernel (mxk<5>, 43 % of a frame): builds scratch/lib_flag_<tag>.so per flag set.
set -e
B=ibl-nerf_amd/build; C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Iinclude -mllvm -amdgpu-mfma-vgpr-form -DIBL_MX_VARIANT=5"
OBJS=$(ls $B/*.o | grep -v "mlp_kernel_mx_trunk_x.o")
build() { tag=$1; shift; /opt/rocm/bin/hipcc $C "$@" -c ibl-nerf_amd/csrc/mlp_kernel_mx.hip -o scratch/flag_$tag.o 2>scratch/flag_$tag.err && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_flag_$tag.so $OBJS scratch/flag_$tag.o && echo "built $tag" || echo "FAILED $tag: $(tail -2 scratch/flag_$tag.err)"; }
build pad50 -mllvm -amdgpu-mfma-padding-ratio=50 &
build pad100 -mllvm -amdgpu-mfma-padding-ratio=100 &
build noigl -mllvm -amdgpu-igrouplp=0 &
build trackers -mllvm -amdgpu-use-amdgpu-trackers=1 &
build o2 -O2 &
build relaxocc -mllvm -amdgpu-schedule-relaxed-occupancy=true &
wait
