// Host-side packer interface (see layout.h for the formats).
#pragma once
#include <stddef.h>
#include "layout.h"

namespace ibl {
size_t blob_floats();  // 798 994: floats in one network's state-dict blob
// blob -> stream_out (STREAM_BYTES) + tab (TAB_FLOATS floats), both host buffers
void pack_network(const float* blob, void* stream_out, float* tab);
// f16 + MX-fp6 variant (layout_mx.h): stream_out holds mx::STREAM_BYTES, tab as above
void pack_network_mx(const float* blob, void* stream_out, float* tab);
}  // namespace ibl
