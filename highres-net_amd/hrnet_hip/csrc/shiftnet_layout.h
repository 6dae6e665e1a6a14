// Packed-parameter layout of ShiftNet (byte offsets into the blob hrn_shiftnet_pack() fills) and the layer table
// (ShiftNet.py:16-47), shared by api.hip and shiftnet_bwd.hip.
#pragma once
#include "common.h"

namespace hrn {

static const int SN_CI[8] = {2, 64, 64, 64, 64, 128, 128, 128};
static const int SN_CO[8] = {64, 64, 64, 64, 128, 128, 128, 128};
static const int SN_POOL[8] = {0, 1, 0, 1, 0, 1, 0, 0};
constexpr int SN_PARTIAL_BLOCKS = 256;

struct SnLayout {
    size_t conv_w[8], conv_b[8], fc1_b, fc2_w, total;        // (fc1.weight, 134 MB, is read in place: not packed)
};
static inline SnLayout sn_layout() {
    SnLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = hrn_align_up(off + bytes, 256); return o; };
    for (int i = 0; i < 8; ++i) { L.conv_w[i] = take((size_t)SN_CI[i] * SN_CO[i] * 9 * 4); L.conv_b[i] = take(SN_CO[i] * 4); }
    L.fc1_b = take(1024 * 4); L.fc2_w = take(2 * 1024 * 4);
    L.total = off;
    return L;
}

}  // namespace hrn
