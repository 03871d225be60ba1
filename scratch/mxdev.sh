#!/bin/bash
# development loop for mlp_kernel_mx.hip: TRUNK-only instantiation (compiles in ~40 s) linked into scratch/lib_mxdev.so
set -e
cd /root/repo
B=ibl-nerf_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DIBL_MX_DEV_TRUNK_ONLY -mllvm -amdgpu-mfma-vgpr-form ${MXFLAGS:-} -c ibl-nerf_amd/csrc/mlp_kernel_mx.hip -o scratch/mx_dev.o -save-temps=obj 2>&1 | grep -v warning || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -x hip -c ibl-nerf_amd/csrc/pack.cpp -o scratch/pack_dev.o 2>&1 | grep -v warning || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_mxdev.so scratch/mx_dev.o $B/mlp_kernel_full.o $B/mlp_kernel_trunk.o $B/mlp_kernel_refl.o $B/mlp_kernel_full_ci.o $B/mlp_kernel_refl_ci.o $B/render_kernels.hip.o $B/pack_kernels.hip.o $B/api.cpp.o scratch/pack_dev.o
S=scratch/mlp_kernel_mx-hip-amdgcn-amd-amdhsa-gfx950.s
echo "uses of m0 outside the DMA asm: $(grep -v "s_mov_b32 m0" $S | grep -c "\bm0\b")"
grep -E "^\s+\.(vgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size)" $S
python3 - <<'PY'
import collections
lines=open('/root/repo/scratch/mlp_kernel_mx-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
idx=[i for i,l in enumerate(lines) if 'v_mfma' in l]
a,b=idx[100],idx[484]
ops=collections.Counter()
for l in lines[a:b]:
    l=l.strip()
    if not l or l[0] in ';.' or l.endswith(':'): continue
    ops[l.split()[0]]+=1
tot=sum(ops.values())
print("instr/MFMA over 2 trunk layers: %.2f" % (tot/384), "  ".join("%s %.2f"%(k.replace('_e32','').replace('_e64',''),v/384) for k,v in ops.most_common(16)))
PY
rm -f scratch/mlp_kernel_mx-h* scratch/mlp_kernel_mx.hip-hip* 
