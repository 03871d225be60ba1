// Host-side packer interface (see layout.h for the formats).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "layout.h"

namespace ibl {
size_t blob_floats();  // 798 994: floats in one network's state-dict blob
// offsets (floats) of layer `layer`'s weight and bias inside the blob; layers in registration order: 0..7 positions_linears,
// 8 views_linears.0, 9 feature_linear, 10 sigma_linear, ... (pack.cpp: LayerId)
void blob_offsets(int layer, size_t* weight, size_t* bias);
// blob -> stream_out (STREAM_BYTES) + tab (TAB_FLOATS floats), both host buffers
void pack_network(const float* blob, void* stream_out, float* tab);
// the same layout with f16 (hi, lo) pairs (mlp_kernel.hip -DIBL_F16X3); not thread-safe (shares the packer's mode flags)
void pack_network_f16x3(const float* blob, void* stream_out, float* tab);
// f16 + MX-fp6 variant (layout_mx.h): stream_out holds mx::STREAM_BYTES, tab as above
void pack_network_mx(const float* blob, void* stream_out, float* tab);
// Gather maps for the device packer: entry = 1 + flat blob index of the weight that lands at that position, 0 = zero.
//   id_stream[STREAM_BYTES/2]: per bf16x3 k-step, the index's low 16 bits sit in the hi-fragment slot and its high bits
//   in the lo-fragment slot; map_mx[(block*64 + lane)*32 + slot]; map_tab[TAB_FLOATS].  Not thread-safe.
void build_pack_maps(std::vector<uint16_t>& id_stream, std::vector<int32_t>& map_mx, std::vector<int32_t>& map_tab);
}  // namespace ibl
