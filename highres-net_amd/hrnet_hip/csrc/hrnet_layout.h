// Packed-parameter layout of HRNet (byte offsets into the blob hrn_hrnet_pack() fills), shared by api.hip and train.hip.
#pragma once
#include <string.h>
#include "../../../include/hrnet_hip.h"
#include "common.h"
#include "conv3x3.h"

namespace hrn {

constexpr size_t ALIGN = 256;

struct HrnetLayout {
    size_t stem_w, stem_b, stem_a;
    size_t enc_w[2 * HRN_MAX_RES_LAYERS], enc_b[2 * HRN_MAX_RES_LAYERS], enc_a[2 * HRN_MAX_RES_LAYERS];
    size_t encf_w, encf_b;
    size_t fres_w[2], fres_b[2], fres_a[2];
    size_t fout_w, fout_b, fout_a;
    size_t dec_w, dec_b, dec_a, fin_w, fin_b;
    size_t total;
};

static inline HrnetLayout hrnet_layout(int dt, int nl) {
    HrnetLayout L;
    memset(&L, 0, sizeof L);
    const size_t es = hrn_esize(dt);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = hrn_align_up(off + bytes, ALIGN); return o; };
    L.stem_w = take(64 * 18 * 4); L.stem_b = take(64 * 4); L.stem_a = take(4);
    for (int i = 0; i < 2 * nl; ++i) { L.enc_w[i] = take(64 * 64 * 9 * es); L.enc_b[i] = take(64 * 4); L.enc_a[i] = take(4); }
    L.encf_w = take(64 * 64 * 9 * es); L.encf_b = take(64 * 4);
    for (int i = 0; i < 2; ++i) { L.fres_w[i] = take(128 * 128 * 9 * es); L.fres_b[i] = take(128 * 4); L.fres_a[i] = take(4); }
    L.fout_w = take(128 * 64 * 9 * es); L.fout_b = take(64 * 4); L.fout_a = take(4);
    L.dec_w = take(64 * 64 * 9 * es); L.dec_b = take(64 * 4); L.dec_a = take(4);
    L.fin_w = take(64 * 4); L.fin_b = take(4);
    L.total = off;
    return L;
}

static inline const unsigned char* at(const void* base, size_t off) { return (const unsigned char*)base + off; }
static inline unsigned char* at(void* base, size_t off) { return (unsigned char*)base + off; }

static inline ConvParams conv_base(int M, int H, int W) {
    ConvParams p;
    memset(&p, 0, sizeof p);
    p.M = M; p.H = H; p.W = W;
    return p;
}

}  // namespace hrn
