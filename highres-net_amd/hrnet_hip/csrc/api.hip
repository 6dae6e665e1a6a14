// extern "C" entry points of libhrnet_hip.so (declared in include/hrnet_hip.h): argument checks, packed-parameter
// and workspace layouts, and the kernel sequences of HRNet.forward / ShiftNet.forward.
#include "../../../include/hrnet_hip.h"
#include "kernels.h"
#include "hrnet_layout.h"
#include "shiftnet_layout.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";
void hrn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

namespace {

using namespace hrn;     // packed-parameter layout, at(), conv_base(): hrnet_layout.h

// ---------------------------------------------------------------- HRNet workspace layout
struct HrnetWs {
    size_t ref, emb, buf_a, buf_b, fused, total;
};
HrnetWs hrnet_ws(int dt, int B, int V, int H, int W) {
    HrnetWs w;
    const size_t es = hrn_esize(dt), hw = (size_t)H * W;
    const size_t stack = (size_t)B * V * hw * 64 * es;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = hrn_align_up(off + bytes, ALIGN); return o; };
    w.ref = take((size_t)B * hw * 4);
    w.emb = take(stack);
    w.buf_a = take(stack);      // encoder ping / fusion t1 (B * V/2 images x 128 ch == stack bytes at most)
    w.buf_b = take(stack);      // encoder pong / fusion t2
    w.fused = take((size_t)B * hw * 64 * es);
    w.total = off;
    return w;
}

// HRN_BF16X3: every activation tensor of the workspace is a pair of bf16 planes; the lo plane of the view stack and of the two
// ping-pong buffers starts half a stack further on (a plane of the fusion's 128-channel intermediates is at most that large), the lo
// plane of the fused state B*HW*64 bf16 further on.  0 for the other dtypes.
size_t stack_lo(int dt, int B, int V, int H, int W) { return dt == HRN_BF16X3 ? (size_t)B * V * H * W * 64 * 2 : 0; }
size_t fused_lo(int dt, int B, int H, int W) { return dt == HRN_BF16X3 ? (size_t)B * H * W * 64 * 2 : 0; }
bool dtype_ok(int dt) { return dt == HRN_F32 || dt == HRN_BF16 || dt == HRN_BF16X3; }

int check_common(int dt, int nl, int B, int V, int H, int W) {
    HRN_CHECK(dtype_ok(dt), -2, "dtype must be HRN_DTYPE_F32, HRN_DTYPE_BF16 or HRN_DTYPE_BF16X3 (got %d)", dt);
    HRN_CHECK(nl >= 0 && nl <= HRN_MAX_RES_LAYERS, -2, "num_layers %d out of range 0..%d", nl, HRN_MAX_RES_LAYERS);
    HRN_CHECK(B > 0 && V > 0 && H > 0 && W > 0, -2, "empty input B=%d V=%d H=%d W=%d", B, V, H, W);
    return 0;
}

int encoder_impl(const void* pk, int dt, int nl, const float* lrs, int B, int V, int H, int W,
                 void* emb, void* ws, const HrnetWs& wl, hipStream_t s) {
    const HrnetLayout L = hrnet_layout(dt, nl);
    const size_t hw = (size_t)H * W;
    float* ref = (float*)at(ws, wl.ref);
    void* bufA = at(ws, wl.buf_a);
    void* bufB = at(ws, wl.buf_b);
    int rc;
    if ((rc = hrn_launch_median(lrs, ref, B, V, H, W, s))) return rc;
    // stem: channel 0 = view, channel 1 = the sample's reference frame; 2->64 + PReLU  (HRNet.py:200-204, :51-53)
    const size_t lo = stack_lo(dt, B, V, H, W);        // bf16x3: lo plane of bufA / bufB / emb (0 otherwise)
    if ((rc = hrn_launch_stem(dt, lrs, hw, ref, V, hw, nullptr, (const float*)at(pk, L.stem_w), (const float*)at(pk, L.stem_b),
                              (const float*)at(pk, L.stem_a), bufA, B * V, H, W, s, lo))) return rc;
    // residual blocks: A -conv+PReLU-> B -conv+PReLU, + A-> A (in place: the residual is read at the stored pixel only)
    for (int l = 0; l < nl; ++l) {
        ConvParams p = conv_base(B * V, H, W);
        p.in = bufA; p.out = bufB; p.in_lo = p.out_lo = lo;
        p.wpk = at(pk, L.enc_w[2 * l]); p.bias = (const float*)at(pk, L.enc_b[2 * l]); p.slope = (const float*)at(pk, L.enc_a[2 * l]);
        if ((rc = hrn_launch_conv3x3(dt, 64, 64, p, s))) return rc;
        ConvParams q = conv_base(B * V, H, W);
        q.in = bufB; q.out = bufA; q.res = bufA; q.res_mode = 1; q.in_lo = q.out_lo = q.res_lo = lo;
        q.wpk = at(pk, L.enc_w[2 * l + 1]); q.bias = (const float*)at(pk, L.enc_b[2 * l + 1]); q.slope = (const float*)at(pk, L.enc_a[2 * l + 1]);
        if ((rc = hrn_launch_conv3x3(dt, 64, 64, q, s))) return rc;
    }
    ConvParams f = conv_base(B * V, H, W);
    f.in = bufA; f.out = emb; f.in_lo = f.out_lo = lo;
    f.wpk = at(pk, L.encf_w); f.bias = (const float*)at(pk, L.encf_b); f.slope = nullptr;
    return hrn_launch_conv3x3(dt, 64, 64, f, s);
}

int fuse_impl(const void* pk, int dt, int nl, int alpha_residual, void* emb, const float* alphas, int B, int V, int H, int W,
              void* fused, void* ws, const HrnetWs& wl, hipStream_t s) {
    const HrnetLayout L = hrnet_layout(dt, nl);
    const size_t hw = (size_t)H * W, es = hrn_esize(dt);
    void* t1 = at(ws, wl.buf_a);
    void* t2 = at(ws, wl.buf_b);
    const size_t lo = stack_lo(dt, B, V, H, W), flo = fused_lo(dt, B, H, W);
    int n = V, rc;
    if (n / 2 == 0) {   // V == 1: no fusion level; mean over one view is the identity (HRNet.py:113,134)
        HRN_HIP(hipMemcpyAsync(fused, emb, (size_t)B * hw * 64 * es, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    while (n / 2 > 0) {
        const int parity = n & 1, half = n >> 1;
        const bool last = (half == 1);
        // g: z = cat(s_i, s_partner) -> t1 = PReLU(conv(z))                 (ResidualBlock first half, HRNet.py:18-19)
        ConvParams a = conv_base(B * half, H, W);
        a.in_pair = 1; a.stack = emb; a.pair_h = half; a.pair_last = n - parity - 1; a.pair_vs = V;
        a.out = t1; a.stack_lo = a.out_lo = lo;
        a.wpk = at(pk, L.fres_w[0]); a.bias = (const float*)at(pk, L.fres_b[0]); a.slope = (const float*)at(pk, L.fres_a[0]);
        if ((rc = hrn_launch_conv3x3(dt, 128, 128, a, s))) return rc;
        // t2 = z + PReLU(conv(t1))                                          (HRNet.py:20-21, :33)
        ConvParams b = conv_base(B * half, H, W);
        b.in = t1; b.out = t2; b.in_lo = b.out_lo = b.stack_lo = lo;
        b.res_mode = 2;      // residual = the same pair gather, straight from the stack
        b.stack = emb; b.pair_h = half; b.pair_last = n - parity - 1; b.pair_vs = V;
        b.wpk = at(pk, L.fres_w[1]); b.bias = (const float*)at(pk, L.fres_b[1]); b.slope = (const float*)at(pk, L.fres_a[1]);
        if ((rc = hrn_launch_conv3x3(dt, 128, 128, b, s))) return rc;
        // f = PReLU(conv(t2)); s_i <- s_i + alpha_partner * f  (or s_i <- f)  (HRNet.py:95-97, :123-128)
        ConvParams c = conv_base(B * half, H, W);
        c.in = t2; c.in_lo = c.res_lo = lo;
        c.out_h = half;
        if (last) { c.out = fused; c.out_vs = 1; c.out_lo = flo; } else { c.out = emb; c.out_vs = V; c.out_lo = lo; }
        c.pair_last = n - parity - 1;
        if (alpha_residual) { c.res_mode = 3; c.res = emb; c.res_vs = V; c.alphas = alphas; c.alpha_vs = V; }
        c.wpk = at(pk, L.fout_w); c.bias = (const float*)at(pk, L.fout_b); c.slope = (const float*)at(pk, L.fout_a);
        if ((rc = hrn_launch_conv3x3(dt, 128, 64, c, s))) return rc;
        n = half;
    }
    return 0;
}

int decoder_impl(const void* pk, int dt, int nl, const void* fused, int N, int H, int W, float* sr, hipStream_t s) {
    const HrnetLayout L = hrnet_layout(dt, nl);
    return hrn_launch_decoder(dt, fused, at(pk, L.dec_w), (const float*)at(pk, L.dec_b), (const float*)at(pk, L.dec_a),
                              (const float*)at(pk, L.fin_w), (const float*)at(pk, L.fin_b), sr, N, H, W, s, fused_lo(dt, N, H, W));
}

// ---------------------------------------------------------------- ShiftNet layouts (packed parameters: shiftnet_layout.h)
struct SnWs {
    size_t means, scale, shift, partial, x, y, fc, xr, fc_partial, total;
};
SnWs sn_ws(int B) {
    SnWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = hrn_align_up(off + bytes, ALIGN); return o; };
    w.means = take((size_t)B * 2 * 4);
    w.scale = take(128 * 4); w.shift = take(128 * 4);
    w.partial = take((size_t)SN_PARTIAL_BLOCKS * 128 * 2 * 8);
    w.x = take((size_t)B * 128 * 128 * 64 * 4);     // conv output (pre-BN), largest at layer 1/2
    w.y = take((size_t)B * 128 * 128 * 64 * 4);     // activation after BN+ReLU(+pool)
    w.fc = take((size_t)B * 1024 * 4);
    w.xr = take((size_t)B * 32768 * 4);             // fc1's input in the reference's flatten order
    w.fc_partial = take(hrn_fc1_partial_bytes());
    w.total = off;
    return w;
}

}  // namespace

// ================================================================= C ABI
extern "C" {

int hrn_version(void) { return HRN_ABI_VERSION; }
const char* hrn_last_error(void) { return g_err; }

size_t hrn_hrnet_packed_bytes(int dtype, int num_layers) {
    if (!dtype_ok(dtype) || num_layers < 0 || num_layers > HRN_MAX_RES_LAYERS) return 0;
    return hrnet_layout(dtype, num_layers).total;
}

int hrn_hrnet_pack(const hrn_hrnet_params* P, int dt, void* packed, size_t packed_bytes, void* stream) {
    HRN_CHECK(P && packed, -2, "hrn_hrnet_pack: null argument");
    int rc;
    if ((rc = check_common(dt, P->num_layers, 1, 1, 1, 1))) return rc;
    const int nl = P->num_layers;
    const HrnetLayout L = hrnet_layout(dt, nl);
    HRN_CHECK(packed_bytes >= L.total, -3, "hrn_hrnet_pack: packed buffer too small (%zu < %zu)", packed_bytes, L.total);
    hipStream_t s = (hipStream_t)stream;
    auto copy = [&](size_t off, const float* src, size_t n) -> int {
        HRN_CHECK(src != nullptr, -2, "hrn_hrnet_pack: null parameter pointer");
        HRN_HIP(hipMemcpyAsync(at(packed, off), src, n * 4, hipMemcpyDeviceToDevice, s));
        return 0;
    };
    if ((rc = copy(L.stem_w, P->enc_init_w, 64 * 18)) || (rc = copy(L.stem_b, P->enc_init_b, 64)) || (rc = copy(L.stem_a, P->enc_init_a, 1))) return rc;
    for (int i = 0; i < 2 * nl; ++i) {
        HRN_CHECK(P->enc_res_w[i], -2, "hrn_hrnet_pack: null encoder weight %d", i);
        if ((rc = hrn_launch_conv_pack(dt, 64, 64, P->enc_res_w[i], at(packed, L.enc_w[i]), s))) return rc;
        if ((rc = copy(L.enc_b[i], P->enc_res_b[i], 64)) || (rc = copy(L.enc_a[i], P->enc_res_a[i], 1))) return rc;
    }
    HRN_CHECK(P->enc_final_w && P->fuse_out_w && P->dec_w, -2, "hrn_hrnet_pack: null weight pointer");
    if ((rc = hrn_launch_conv_pack(dt, 64, 64, P->enc_final_w, at(packed, L.encf_w), s))) return rc;
    if ((rc = copy(L.encf_b, P->enc_final_b, 64))) return rc;
    for (int i = 0; i < 2; ++i) {
        HRN_CHECK(P->fuse_res_w[i], -2, "hrn_hrnet_pack: null fuse weight %d", i);
        if ((rc = hrn_launch_conv_pack(dt, 128, 128, P->fuse_res_w[i], at(packed, L.fres_w[i]), s))) return rc;
        if ((rc = copy(L.fres_b[i], P->fuse_res_b[i], 128)) || (rc = copy(L.fres_a[i], P->fuse_res_a[i], 1))) return rc;
    }
    if ((rc = hrn_launch_conv_pack(dt, 128, 64, P->fuse_out_w, at(packed, L.fout_w), s))) return rc;
    if ((rc = copy(L.fout_b, P->fuse_out_b, 64)) || (rc = copy(L.fout_a, P->fuse_out_a, 1))) return rc;
    if ((rc = hrn_launch_decoder_pack(dt == HRN_BF16X3 ? HRN_F32 : dt, P->dec_w, at(packed, L.dec_w), s))) return rc;     // bf16x3: the decoder is the fp32 one
    if ((rc = copy(L.dec_b, P->dec_b, 64)) || (rc = copy(L.dec_a, P->dec_a, 1))) return rc;
    if ((rc = copy(L.fin_w, P->fin_w, 64)) || (rc = copy(L.fin_b, P->fin_b, 1))) return rc;
    return 0;
}

size_t hrn_hrnet_workspace_bytes(int dtype, int B, int V, int H, int W) {
    if (!dtype_ok(dtype) || B <= 0 || V <= 0 || H <= 0 || W <= 0) return 0;
    return hrnet_ws(dtype, B, V, H, W).total;
}

int hrn_encoder_forward(const void* packed, int dt, int nl, const float* lrs, int B, int V, int H, int W,
                        void* emb, void* ws, size_t ws_bytes, void* stream) {
    int rc;
    if ((rc = check_common(dt, nl, B, V, H, W))) return rc;
    HRN_CHECK(packed && lrs && emb && ws, -2, "hrn_encoder_forward: null argument");
    const HrnetWs wl = hrnet_ws(dt, B, V, H, W);
    HRN_CHECK(ws_bytes >= wl.total, -3, "hrn_encoder_forward: workspace too small (%zu < %zu)", ws_bytes, wl.total);
    return encoder_impl(packed, dt, nl, lrs, B, V, H, W, emb, ws, wl, (hipStream_t)stream);
}

int hrn_fuse_forward(const void* packed, int dt, int nl, int alpha_residual, void* emb, const float* alphas,
                     int B, int V, int H, int W, void* fused, void* ws, size_t ws_bytes, void* stream) {
    int rc;
    if ((rc = check_common(dt, nl, B, V, H, W))) return rc;
    HRN_CHECK(packed && emb && alphas && fused && ws, -2, "hrn_fuse_forward: null argument");
    const HrnetWs wl = hrnet_ws(dt, B, V, H, W);
    HRN_CHECK(ws_bytes >= wl.total, -3, "hrn_fuse_forward: workspace too small (%zu < %zu)", ws_bytes, wl.total);
    return fuse_impl(packed, dt, nl, alpha_residual, emb, alphas, B, V, H, W, fused, ws, wl, (hipStream_t)stream);
}

int hrn_decoder_forward(const void* packed, int dt, int nl, const void* fused, int N, int H, int W, float* sr, void* stream) {
    int rc;
    if ((rc = check_common(dt, nl, N, 1, H, W))) return rc;
    HRN_CHECK(packed && fused && sr, -2, "hrn_decoder_forward: null argument");
    return decoder_impl(packed, dt, nl, fused, N, H, W, sr, (hipStream_t)stream);
}

int hrn_hrnet_forward(const void* packed, int dt, int nl, int alpha_residual, const float* lrs, const float* alphas,
                      int B, int V, int H, int W, float* sr, void* ws, size_t ws_bytes, void* stream) {
    int rc;
    if ((rc = check_common(dt, nl, B, V, H, W))) return rc;
    HRN_CHECK(packed && lrs && alphas && sr && ws, -2, "hrn_hrnet_forward: null argument");
    const HrnetWs wl = hrnet_ws(dt, B, V, H, W);
    HRN_CHECK(ws_bytes >= wl.total, -3, "hrn_hrnet_forward: workspace too small (%zu < %zu)", ws_bytes, wl.total);
    hipStream_t s = (hipStream_t)stream;
    void* emb = at(ws, wl.emb);
    void* fused = at(ws, wl.fused);
    if ((rc = encoder_impl(packed, dt, nl, lrs, B, V, H, W, emb, ws, wl, s))) return rc;
    if ((rc = fuse_impl(packed, dt, nl, alpha_residual, emb, alphas, B, V, H, W, fused, ws, wl, s))) return rc;
    return decoder_impl(packed, dt, nl, fused, B, H, W, sr, s);
}

// ----------------------------------------------------------------- ShiftNet
size_t hrn_shiftnet_packed_bytes(void) { return sn_layout().total; }

int hrn_shiftnet_pack(const hrn_shiftnet_params* P, void* packed, size_t packed_bytes, void* stream) {
    HRN_CHECK(P && packed, -2, "hrn_shiftnet_pack: null argument");
    const SnLayout L = sn_layout();
    HRN_CHECK(packed_bytes >= L.total, -3, "hrn_shiftnet_pack: packed buffer too small (%zu < %zu)", packed_bytes, L.total);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    for (int i = 0; i < 8; ++i) {
        HRN_CHECK(P->conv_w[i] && P->conv_b[i], -2, "hrn_shiftnet_pack: null conv parameter %d", i);
        if (i == 0) {
            HRN_HIP(hipMemcpyAsync(at(packed, L.conv_w[0]), P->conv_w[0], 64 * 18 * 4, hipMemcpyDeviceToDevice, s));
        } else if ((rc = hrn_launch_conv_pack(HRN_F32, SN_CI[i], SN_CO[i], P->conv_w[i], at(packed, L.conv_w[i]), s))) {
            return rc;
        }
        HRN_HIP(hipMemcpyAsync(at(packed, L.conv_b[i]), P->conv_b[i], SN_CO[i] * 4, hipMemcpyDeviceToDevice, s));
    }
    HRN_CHECK(P->fc1_b && P->fc2_w, -2, "hrn_shiftnet_pack: null fc parameter");
    HRN_HIP(hipMemcpyAsync(at(packed, L.fc1_b), P->fc1_b, 1024 * 4, hipMemcpyDeviceToDevice, s));
    HRN_HIP(hipMemcpyAsync(at(packed, L.fc2_w), P->fc2_w, 2 * 1024 * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

size_t hrn_shiftnet_workspace_bytes(int B) { return B > 0 ? sn_ws(B).total : 0; }

int hrn_shiftnet_forward(const void* packed, const hrn_shiftnet_params* P, const float* x, int B, int train_bn, float momentum,
                         const unsigned char* dropout_mask, float* theta, void* ws, size_t ws_bytes, void* stream) {
    HRN_CHECK(packed && P && x && theta && ws, -2, "hrn_shiftnet_forward: null argument");
    HRN_CHECK(P->fc1_w, -2, "hrn_shiftnet_forward: params->fc1_w is null (fc1.weight is read in place)");
    HRN_CHECK(B > 0, -2, "hrn_shiftnet_forward: empty batch");
    const SnLayout L = sn_layout();
    const SnWs wl = sn_ws(B);
    HRN_CHECK(ws_bytes >= wl.total, -3, "hrn_shiftnet_forward: workspace too small (%zu < %zu)", ws_bytes, wl.total);
    hipStream_t s = (hipStream_t)stream;
    float* means = (float*)at(ws, wl.means);
    float* scale = (float*)at(ws, wl.scale);
    float* shift = (float*)at(ws, wl.shift);
    double* partial = (double*)at(ws, wl.partial);
    float* bx = (float*)at(ws, wl.x);
    float* by = (float*)at(ws, wl.y);
    float* fc = (float*)at(ws, wl.fc);
    int rc, hsz = 128;
    const size_t plane = 128 * 128;
    if ((rc = hrn_launch_plane_mean(x, means, B * 2, plane, s))) return rc;                      // ShiftNet.py:58
    for (int i = 0; i < 8; ++i) {
        HRN_CHECK(P->bn_g[i] && P->bn_b[i] && P->bn_rm[i] && P->bn_rv[i], -2, "hrn_shiftnet_forward: null BatchNorm tensor %d", i);
        const int C = SN_CO[i];
        if (i > 0 && !train_bn) {
            // eval mode: BatchNorm (running statistics) + ReLU are the convolution's epilogue - one launch per layer instead of three and
            // no pre-BatchNorm tensor; the pooled layers keep a pool-only pass                         ShiftNet.py:16-42 in .eval()
            if ((rc = hrn_launch_bn_fold(P->bn_g[i], P->bn_b[i], P->bn_rm[i], P->bn_rv[i], 1e-5f, (const float*)at(packed, L.conv_b[i]),
                                         scale, shift, C, s))) return rc;
            ConvParams p = conv_base(B, hsz, hsz);
            p.in = by; p.out = bx;
            p.wpk = at(packed, L.conv_w[i]); p.scale = scale; p.bias = shift; p.relu = 1;
            if ((rc = hrn_launch_conv3x3(HRN_F32, SN_CI[i], C, p, s))) return rc;
            if (SN_POOL[i]) {
                if ((rc = hrn_launch_bn_act_pool(bx, nullptr, nullptr, by, B, hsz, hsz, C, 1, s))) return rc;
                hsz /= 2;
            } else {
                float* t = bx; bx = by; by = t;             // the activation is where the conv wrote it
            }
            continue;
        }
        if (i == 0) {
            if ((rc = hrn_launch_stem(HRN_F32, x, 2 * plane, x + plane, 1, 2 * plane, means, (const float*)at(packed, L.conv_w[0]),
                                      (const float*)at(packed, L.conv_b[0]), nullptr, bx, B, hsz, hsz, s))) return rc;
        } else {
            ConvParams p = conv_base(B, hsz, hsz);
            p.in = by; p.out = bx;
            p.wpk = at(packed, L.conv_w[i]); p.bias = (const float*)at(packed, L.conv_b[i]);
            if ((rc = hrn_launch_conv3x3(HRN_F32, SN_CI[i], SN_CO[i], p, s))) return rc;
        }
        if (train_bn) {
            if ((rc = hrn_launch_bn_stats(bx, (size_t)B * hsz * hsz, C, P->bn_g[i], P->bn_b[i], 1e-5f, scale, shift,
                                          P->bn_rm[i], P->bn_rv[i], momentum, partial, SN_PARTIAL_BLOCKS, s))) return rc;
        } else {
            if ((rc = hrn_launch_bn_fold(P->bn_g[i], P->bn_b[i], P->bn_rm[i], P->bn_rv[i], 1e-5f, nullptr, scale, shift, C, s))) return rc;
        }
        if ((rc = hrn_launch_bn_act_pool(bx, scale, shift, by, B, hsz, hsz, C, SN_POOL[i], s))) return rc;
        if (SN_POOL[i]) hsz /= 2;
    }
    // by: [B][16][16][128] NHWC -> xr [B][c*256 + hw], the reference's flatten order (dropout folded in); fc1.weight is read in place
    float* xr = (float*)at(ws, wl.xr);
    if ((rc = hrn_launch_fc_to_ref(by, dropout_mask, xr, B, s))) return rc;
    if ((rc = hrn_launch_fc1(xr, P->fc1_w, (const float*)at(packed, L.fc1_b), fc, B, (float*)at(ws, wl.fc_partial), s))) return rc;
    return hrn_launch_fc2(fc, (const float*)at(packed, L.fc2_w), theta, B, s);
}

// ----------------------------------------------------------------- Lanczos
int hrn_lanczos_kernel(const float* dx, int n, float* taps, void* stream) {
    HRN_CHECK(n >= 0 && (n == 0 || (dx && taps)), -2, "hrn_lanczos_kernel: bad argument");
    return hrn_launch_lanczos_taps(dx, n, taps, (hipStream_t)stream);
}

int hrn_lanczos_shift(const float* img, const float* shift, int b, int c, int H, int W, float* out, void* stream) {
    HRN_CHECK(b >= 0 && c >= 0 && H > 0 && W > 0, -2, "hrn_lanczos_shift: bad shape");
    HRN_CHECK(b * c == 0 || (img && shift && out), -2, "hrn_lanczos_shift: null argument");
    return hrn_launch_lanczos_shift(img, shift, b, c, H, W, out, (hipStream_t)stream);
}

// ----------------------------------------------------------------- loss / score reductions
size_t hrn_lanczos_shift_backward_workspace_bytes(int b, int c, int H, int W) {
    if (b <= 0 || c <= 0 || H <= 0 || W <= 0) return 0;
    return hrn_lanczos_bwd_workspace_bytes_impl(b, c, H, W);
}

int hrn_lanczos_shift_backward(const float* img, const float* shift, const float* d_out, int b, int c, int H, int W, float* d_img,
                               float* d_shift, void* ws, size_t ws_bytes, void* stream) {
    HRN_CHECK(img && shift && d_out && ws, -2, "hrn_lanczos_shift_backward: null argument");
    HRN_CHECK(b > 0 && c > 0, -2, "hrn_lanczos_shift_backward: empty input b=%d c=%d", b, c);
    HRN_CHECK(ws_bytes >= hrn_lanczos_bwd_workspace_bytes_impl(b, c, H, W), -3, "hrn_lanczos_shift_backward: workspace too small");
    return hrn_launch_lanczos_shift_bwd(img, shift, d_out, b, c, H, W, d_img, d_shift, ws, (hipStream_t)stream);
}

int hrn_get_loss(const float* srs, const float* hrs, const float* maps, int B, int S, int crop, int metric, float* out, void* stream) {
    HRN_CHECK(B > 0 && S > 0 && crop >= 0 && 2 * crop < S, -2, "hrn_get_loss: bad shape B=%d S=%d crop=%d", B, S, crop);
    HRN_CHECK(metric >= 0 && metric <= 2, -2, "hrn_get_loss: metric must be 0 (masked_MSE), 1 (cMSE) or 2 (cPSNR)");
    HRN_CHECK(srs && hrs && maps && out, -2, "hrn_get_loss: null argument");
    return hrn_launch_masked_cmse(srs, hrs, maps, B, S, crop, metric, out, (hipStream_t)stream);
}

size_t hrn_get_loss_train_workspace_bytes(int B) { return B > 0 ? hrn_loss_train_workspace_bytes_impl(B) : 0; }

int hrn_get_loss_train(const float* srs, const float* hrs, const float* maps, int B, int S, int crop, int metric, float* out,
                       double* stats, void* ws, size_t ws_bytes, void* stream) {
    HRN_CHECK(B > 0 && B <= 65535 && S > 0 && crop >= 0 && 2 * crop < S, -2, "hrn_get_loss_train: bad shape B=%d S=%d crop=%d", B, S, crop);
    HRN_CHECK(metric == 1 || metric == 2, -2, "hrn_get_loss_train: metric must be 1 (cMSE) or 2 (cPSNR); masked_MSE has no registered form");
    HRN_CHECK(srs && hrs && maps && out && stats && ws, -2, "hrn_get_loss_train: null argument");
    HRN_CHECK(ws_bytes >= hrn_loss_train_workspace_bytes_impl(B), -3, "hrn_get_loss_train: workspace too small");
    return hrn_launch_loss_train(srs, hrs, maps, B, S, crop, metric, out, stats, (double*)ws, (hipStream_t)stream);
}

int hrn_get_loss_backward(const float* srs, const float* hrs, const float* maps, const double* stats, const float* d_out, int B, int S,
                          int crop, int metric, float* d_srs, void* stream) {
    HRN_CHECK(B > 0 && B <= 65535 && S > 0 && crop >= 0 && 2 * crop < S, -2, "hrn_get_loss_backward: bad shape B=%d S=%d crop=%d", B, S, crop);
    HRN_CHECK(metric == 1 || metric == 2, -2, "hrn_get_loss_backward: metric must be 1 (cMSE) or 2 (cPSNR)");
    HRN_CHECK(srs && hrs && maps && stats && d_out && d_srs, -2, "hrn_get_loss_backward: null argument");
    return hrn_launch_loss_backward(srs, hrs, maps, stats, d_out, B, S, crop, metric, d_srs, (hipStream_t)stream);
}

size_t hrn_shift_cpsnr_workspace_bytes(int B, int border) {
    if (B <= 0 || border < 0) return 0;
    return (size_t)B * (2 * border + 1) * (2 * border + 1) * sizeof(double);
}

int hrn_shift_cpsnr(const float* srs, const float* hrs, const float* maps, int B, int S, int border, int clip, float* out,
                    void* ws, size_t ws_bytes, void* stream) {
    HRN_CHECK(B > 0 && border >= 0 && S > 2 * border, -2, "hrn_shift_cpsnr: bad shape B=%d S=%d border=%d", B, S, border);
    HRN_CHECK(B <= 65535, -2, "hrn_shift_cpsnr: batch %d exceeds the grid limit", B);
    HRN_CHECK(srs && hrs && maps && out && ws, -2, "hrn_shift_cpsnr: null argument");
    HRN_CHECK(ws_bytes >= hrn_shift_cpsnr_workspace_bytes(B, border), -3, "hrn_shift_cpsnr: workspace too small");
    return hrn_launch_shift_cpsnr(srs, hrs, maps, B, S, border, clip, (double*)ws, out, (hipStream_t)stream);
}

}  // extern "C"
