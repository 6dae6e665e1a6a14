// Training path of HRNet (fp32, or bf16x3: every activation / gradient tensor a pair of bf16 planes, three bf16 MFMAs per product - the
// convolutions, data gradients and weight gradients then run on conv3x3_v6x3.hip / wgrad_x3.hip, the elementwise passes on the same
// kernels templated on the storage): a forward that keeps what the backward needs, and the backward itself
// (`srs = fusion_model(lrs, alphas)` ... `loss.backward()`, src/train.py:172-190).
//
// Forward (HRNet.py:186-211) is the inference kernel sequence with every intermediate kept in the training workspace:
//   encoder   a0 = PReLU(stem);  h_l = PReLU(conv(a_l));  r_l = PReLU(conv(h_l));  a_{l+1} = a_l + r_l;  stack_0 = conv(a_nl)
//   level     z = cat(s_i, s_partner);  t1 = PReLU(convA(z));  u = PReLU(convB(t1));  t2 = z + u;  f = PReLU(convC(t2));
//             stack_{l+1}[i] = s_i + alpha_partner f   (i < n/2)
//   decoder   sr = conv1x1(PReLU(deconv(stack_T)))
// Backward walks it in reverse with three primitives per convolution: PReLU backward on the stored post-activation
// (+ slope gradient), the weight/bias gradient (backward.hip), and the data gradient, which is the forward convolution
// kernel run on the transposed, tap-flipped weights.  All gradients are accumulated (+=) into the caller's buffers.
#include "../../../include/hrnet_hip.h"
#include "kernels.h"
#include "backward.h"
#include "hrnet_layout.h"

using namespace hrn;

namespace {

constexpr int TMAX = 16;

struct TrainWs {
    int T;                                  // fusion levels
    int n_in[TMAX + 1];                     // views entering level l (n_in[T] = views left at the end)
    size_t ref, a[HRN_MAX_RES_LAYERS + 1], h[HRN_MAX_RES_LAYERS], r[HRN_MAX_RES_LAYERS];
    size_t stack[TMAX + 1], t1[TMAX], u[TMAX], t2[TMAX], f[TMAX];
    size_t g[5];                            // backward: five gradient buffers of one full activation each
    size_t xpre;                            // backward: a recomputed pre-activation (used only behind a PReLU whose slope is <= 0)
    size_t wt, wtp, zero_bias, scratch;
    size_t dec_f, dec_g;                    // bf16x3: f32 copies of the fused state and of its gradient (the decoder's backward is the fp32 kernel)
    size_t total;
};

// bf16x3: a tensor of n elements is a pair of bf16 planes, the lo plane 2 n bytes behind the hi plane (0 for fp32)
inline size_t lo_of(int dt, size_t n) { return dt == HRN_BF16X3 ? n * 2 : 0; }

int num_cus() { return hrn_device_cus(); }

TrainWs train_ws(int nl, int B, int V, int H, int W) {
    TrainWs w;
    memset(&w, 0, sizeof w);
    const size_t hw = (size_t)H * W;
    const size_t S = (size_t)B * V * hw * 64 * 4;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = hrn_align_up(off + bytes, ALIGN); return o; };
    w.ref = take((size_t)B * hw * 4);
    for (int l = 0; l <= nl; ++l) w.a[l] = take(S);
    for (int l = 0; l < nl; ++l) { w.h[l] = take(S); w.r[l] = take(S); }
    int n = V, T = 0;
    w.n_in[0] = V;
    w.stack[0] = take(S);
    while (n / 2 > 0 && T < TMAX) {
        const int half = n / 2;
        const size_t s128 = (size_t)B * half * hw * 128 * 4, s64 = (size_t)B * half * hw * 64 * 4;
        w.t1[T] = take(s128); w.u[T] = take(s128); w.t2[T] = take(s128); w.f[T] = take(s64);
        w.stack[T + 1] = take(s64);
        n = half;
        ++T;
        w.n_in[T] = n;
    }
    w.T = T;
    for (int i = 0; i < 5; ++i) w.g[i] = take(S);
    w.xpre = take(S);
    w.wt = take((size_t)128 * 128 * 9 * 4);
    w.wtp = take((size_t)128 * 128 * 9 * 4);
    w.zero_bias = take(128 * 4);
    size_t sc = hrn_bwd_scratch_bytes(num_cus());
    const size_t sd = hrn_decoder_bwd_scratch_bytes(num_cus());
    if (sd > sc) sc = sd;
    w.scratch = take(sc);
    w.dec_f = take((size_t)B * hw * 64 * 4);
    w.dec_g = take((size_t)B * hw * 64 * 4);
    w.total = off;
    return w;
}

int check_train(int nl, int B, int V, int H, int W) {
    HRN_CHECK(nl >= 0 && nl <= HRN_MAX_RES_LAYERS, -2, "num_layers %d out of range 0..%d", nl, HRN_MAX_RES_LAYERS);
    HRN_CHECK(B > 0 && V > 0 && H > 0 && W > 0, -2, "empty input B=%d V=%d H=%d W=%d", B, V, H, W);
    HRN_CHECK(V < (1 << TMAX), -2, "too many views (%d)", V);
    return 0;
}

// y = conv3x3(x) (+ PReLU) on the forward f32 kernel
int conv_fwd(int dt, int cin, int cout, const void* x, void* y, const void* wpk, const float* bias, const float* slope, int M, int H, int W,
             hipStream_t s) {
    ConvParams p = conv_base(M, H, W);
    p.in = x; p.out = y; p.wpk = wpk; p.bias = bias; p.slope = slope;
    p.in_lo = lo_of(dt, (size_t)M * H * W * cin); p.out_lo = lo_of(dt, (size_t)M * H * W * cout);
    return hrn_launch_conv3x3(dt, cin, cout, p, s);
}

// dx = conv3x3(g, W^T flipped) (+ res): the data gradient of a cin -> cout convolution with raw weights w [cout][cin][3][3]
int conv_dgrad(int dt, int cin, int cout, const float* w, const float* g, float* dx, const float* res, int M, int H, int W, void* tws,
               const TrainWs& L, hipStream_t s) {
    return hrn_conv_dgrad(cin, cout, w, g, dx, res, M, H, W, (float*)at(tws, L.wt), at(tws, L.wtp), (const float*)at(tws, L.zero_bias), s, dt);
}

// dW += the weight gradient of a cin -> cout convolution: x plain [M][H][W][cin], or (x == nullptr) the pair gather of `stack`
int conv_wgrad(int dt, const float* x, const float* stack, int pair_h, int pair_last, int pair_vs, int Bn, const float* g, int M, int H, int W,
               int cin, int cout, float* dw, void* sc, int cus, hipStream_t s) {
    if (dt != HRN_BF16X3) return hrn_launch_conv_wgrad(x, stack, x ? 0 : 1, pair_h, pair_last, pair_vs, g, M, H, W, cin, cout, dw, sc, cus, s);
    const size_t hw = (size_t)H * W;
    const size_t x_lo = x ? (size_t)M * hw * cin * 2 : (size_t)Bn * pair_vs * hw * 64 * 2;     // the stack of the level: Bn samples x pair_vs views
    return hrn_launch_conv_wgrad_x3(x, stack, x_lo, x ? 0 : 1, pair_h, pair_last, pair_vs, g, (size_t)M * hw * cout * 2, M, H, W, cin, cout, dw, sc,
                                    cus, s);
}

// z + u for the pair gather z of a level: t2[b*half + i][p][c] = (c < 64 ? s_i : s_partner)[p][c % 64] + u[...]
template <bool X3>
__global__ __launch_bounds__(256) void pair_add_kernel(const void* __restrict__ stack, int n_in, int half, int pair_last,
                                                       const void* __restrict__ u, void* __restrict__ t2, size_t hw, int B) {
    const size_t total = (size_t)B * half * hw * 32;            // float4 units, 32 per pixel
    const size_t lo_s = (size_t)B * n_in * hw * 128, lo_u = total * 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pixg = i >> 5;
        const int part = (int)(i & 31);
        const size_t img = pixg / hw, pix = pixg - img * hw;
        const int b = (int)(img / half), v = (int)(img - (size_t)b * half);
        const int src = part < 16 ? v : pair_last - v;
        const f32x4 z = act_ld4<X3>(stack, lo_s, (((size_t)b * n_in + src) * hw + pix) * 16 + (part & 15));
        act_st4<X3>(t2, lo_u, i, z + act_ld4<X3>(u, lo_u, i));
    }
}

}  // namespace

extern "C" {

size_t hrn_hrnet_train_workspace_bytes(int num_layers, int B, int V, int H, int W) {
    if (num_layers < 0 || num_layers > HRN_MAX_RES_LAYERS || B <= 0 || V <= 0 || H <= 0 || W <= 0 || V >= (1 << TMAX)) return 0;
    return train_ws(num_layers, B, V, H, W).total;
}

int hrn_hrnet_forward_train(const void* pk, int nl, int alpha_residual, const float* lrs, const float* alphas, int B, int V, int H, int W,
                            float* sr, void* tws, size_t tws_bytes, void* stream) {
    return hrn_hrnet_forward_train_dt(pk, HRN_F32, nl, alpha_residual, lrs, alphas, B, V, H, W, sr, tws, tws_bytes, stream);
}

int hrn_hrnet_backward(const void* pk, const hrn_hrnet_params* Pr, int alpha_residual, const float* lrs, const float* alphas, int B, int V,
                       int H, int W, const float* d_sr, const hrn_hrnet_params* Gr, void* tws, size_t tws_bytes, void* stream) {
    return hrn_hrnet_backward_dt(pk, HRN_F32, Pr, alpha_residual, lrs, alphas, B, V, H, W, d_sr, Gr, tws, tws_bytes, stream);
}

int hrn_hrnet_forward_train_dt(const void* pk, int dt, int nl, int alpha_residual, const float* lrs, const float* alphas, int B, int V, int H,
                               int W, float* sr, void* tws, size_t tws_bytes, void* stream) {
    int rc;
    if ((rc = check_train(nl, B, V, H, W))) return rc;
    HRN_CHECK(dt == HRN_F32 || dt == HRN_BF16X3, -2, "hrn_hrnet_forward_train: dtype must be HRN_DTYPE_F32 or HRN_DTYPE_BF16X3 (got %d)", dt);
    HRN_CHECK(pk && lrs && alphas && sr && tws, -2, "hrn_hrnet_forward_train: null argument");
    const TrainWs L = train_ws(nl, B, V, H, W);
    HRN_CHECK(tws_bytes >= L.total, -3, "hrn_hrnet_forward_train: workspace too small (%zu < %zu)", tws_bytes, L.total);
    const HrnetLayout P = hrnet_layout(dt, nl);
    hipStream_t s = (hipStream_t)stream;
    const size_t hw = (size_t)H * W;
    const int M = B * V;
    float* ref = (float*)at(tws, L.ref);
    if ((rc = hrn_launch_median(lrs, ref, B, V, H, W, s))) return rc;
    if ((rc = hrn_launch_stem(dt, lrs, hw, ref, V, hw, nullptr, (const float*)at(pk, P.stem_w), (const float*)at(pk, P.stem_b),
                              (const float*)at(pk, P.stem_a), at(tws, L.a[0]), M, H, W, s, lo_of(dt, (size_t)M * hw * 64)))) return rc;
    for (int l = 0; l < nl; ++l) {
        if ((rc = conv_fwd(dt, 64, 64, at(tws, L.a[l]), at(tws, L.h[l]), at(pk, P.enc_w[2 * l]), (const float*)at(pk, P.enc_b[2 * l]),
                           (const float*)at(pk, P.enc_a[2 * l]), M, H, W, s))) return rc;
        if ((rc = conv_fwd(dt, 64, 64, at(tws, L.h[l]), at(tws, L.r[l]), at(pk, P.enc_w[2 * l + 1]), (const float*)at(pk, P.enc_b[2 * l + 1]),
                           (const float*)at(pk, P.enc_a[2 * l + 1]), M, H, W, s))) return rc;
        if ((rc = hrn_launch_add((const float*)at(tws, L.a[l]), (const float*)at(tws, L.r[l]), (float*)at(tws, L.a[l + 1]),
                                 (size_t)M * hw * 64, s, dt))) return rc;
    }
    if ((rc = conv_fwd(dt, 64, 64, at(tws, L.a[nl]), at(tws, L.stack[0]), at(pk, P.encf_w), (const float*)at(pk, P.encf_b), nullptr, M, H, W, s)))
        return rc;
    for (int t = 0; t < L.T; ++t) {
        const int n = L.n_in[t], half = n / 2, pair_last = n - (n & 1) - 1;
        const float* st = (const float*)at(tws, L.stack[t]);
        ConvParams a = conv_base(B * half, H, W);
        a.in_pair = 1; a.stack = st; a.pair_h = half; a.pair_last = pair_last; a.pair_vs = n;
        a.out = at(tws, L.t1[t]);
        a.stack_lo = lo_of(dt, (size_t)B * n * hw * 64); a.out_lo = lo_of(dt, (size_t)B * half * hw * 128);
        a.wpk = at(pk, P.fres_w[0]); a.bias = (const float*)at(pk, P.fres_b[0]); a.slope = (const float*)at(pk, P.fres_a[0]);
        if ((rc = hrn_launch_conv3x3(dt, 128, 128, a, s))) return rc;
        if ((rc = conv_fwd(dt, 128, 128, at(tws, L.t1[t]), at(tws, L.u[t]), at(pk, P.fres_w[1]), (const float*)at(pk, P.fres_b[1]),
                           (const float*)at(pk, P.fres_a[1]), B * half, H, W, s))) return rc;
        {
            const size_t total4 = (size_t)B * half * hw * 32;
            size_t grid = (total4 + 255) / 256;
            if (grid > 4096) grid = 4096;
            if (dt == HRN_BF16X3) hipLaunchKernelGGL(pair_add_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, (const void*)st, n, half, pair_last, (const void*)at(tws, L.u[t]),
                                                     (void*)at(tws, L.t2[t]), hw, B);
            else hipLaunchKernelGGL(pair_add_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, (const void*)st, n, half, pair_last, (const void*)at(tws, L.u[t]),
                                    (void*)at(tws, L.t2[t]), hw, B);
            HRN_LAUNCH_CHECK();
        }
        if ((rc = conv_fwd(dt, 128, 64, at(tws, L.t2[t]), at(tws, L.f[t]), at(pk, P.fout_w), (const float*)at(pk, P.fout_b),
                           (const float*)at(pk, P.fout_a), B * half, H, W, s))) return rc;
        if ((rc = hrn_launch_fuse_update(st, n, (const float*)at(tws, L.f[t]), alphas, V, pair_last, half, alpha_residual,
                                         (float*)at(tws, L.stack[t + 1]), hw, B, s, dt))) return rc;
    }
    // views left after the last level: 1 (or V itself for V == 1); torch.mean over them (HRNet.py:134) is the identity
    return hrn_launch_decoder(dt, at(tws, L.stack[L.T]), at(pk, P.dec_w), (const float*)at(pk, P.dec_b), (const float*)at(pk, P.dec_a),
                              (const float*)at(pk, P.fin_w), (const float*)at(pk, P.fin_b), sr, B, H, W, s,
                              lo_of(dt, (size_t)B * L.n_in[L.T] * hw * 64));
}

int hrn_hrnet_backward_dt(const void* pk, int dt, const hrn_hrnet_params* Pr, int alpha_residual, const float* lrs, const float* alphas, int B,
                          int V, int H, int W, const float* d_sr, const hrn_hrnet_params* Gr, void* tws, size_t tws_bytes, void* stream) {
    int rc;
    HRN_CHECK(dt == HRN_F32 || dt == HRN_BF16X3, -2, "hrn_hrnet_backward: dtype must be HRN_DTYPE_F32 or HRN_DTYPE_BF16X3 (got %d)", dt);
    HRN_CHECK(pk && Pr && Gr && lrs && alphas && d_sr && tws, -2, "hrn_hrnet_backward: null argument");
    const int nl = Pr->num_layers;
    if ((rc = check_train(nl, B, V, H, W))) return rc;
    const TrainWs L = train_ws(nl, B, V, H, W);
    HRN_CHECK(tws_bytes >= L.total, -3, "hrn_hrnet_backward: workspace too small (%zu < %zu)", tws_bytes, L.total);
    hipStream_t s = (hipStream_t)stream;
    const size_t hw = (size_t)H * W;
    const int M = B * V, cus = num_cus();
    void* sc = at(tws, L.scratch);
    float* G[5];
    for (int i = 0; i < 5; ++i) G[i] = (float*)at(tws, L.g[i]);
    HRN_HIP(hipMemsetAsync(at(tws, L.zero_bias), 0, 128 * 4, s));
    // gradients are handed over as mutable buffers in a params-shaped struct
    auto mut = [](const float* p) { return const_cast<float*>(p); };
    // PReLU backward works from the stored post-activation while the slope is positive.  For a slope <= 0 (the reference allows any)
    // the pre-activation is recomputed into `xpre` by the forward kernel without activation - a launch that does nothing unless the
    // slope on the device says so (ConvParams::only_if_nonpos): no host round trip, ~3 us per PReLU in the usual case.
    const HrnetLayout P = hrnet_layout(dt, nl);
    float* xpre = (float*)at(tws, L.xpre);
    auto pre = [&](int cin, int cout, const void* x, const void* wpk, const float* bias, const float* slope, int Mi) -> int {
        ConvParams q = conv_base(Mi, H, W);
        q.in = x; q.out = xpre; q.wpk = wpk; q.bias = bias; q.only_if_nonpos = slope;
        q.in_lo = lo_of(dt, (size_t)Mi * hw * cin); q.out_lo = lo_of(dt, (size_t)Mi * hw * cout);
        return hrn_launch_conv3x3(dt, cin, cout, q, s);
    };

    // ---- decoder: d_sr -> d stack_T (one view left)                                  HRNet.py:147-156,167-169
    float* dsn = G[0];                      // gradient of the views leaving the current level
    if (dt == HRN_BF16X3) {
        // the decoder's backward is the fp32 kernel (33 MB of state at the training shape): the fused state as f32, its gradient back as planes
        const size_t nf = (size_t)B * L.n_in[L.T] * hw * 64;
        float* ff = (float*)at(tws, L.dec_f);
        float* fg = (float*)at(tws, L.dec_g);
        if ((rc = hrn_launch_planes_to_f32(at(tws, L.stack[L.T]), nf * 2, ff, nf, s))) return rc;
        if ((rc = hrn_launch_decoder_bwd(ff, d_sr, Pr->dec_w, Pr->dec_b, Pr->dec_a, Pr->fin_w, fg,
                                         mut(Gr->dec_w), mut(Gr->dec_b), mut(Gr->dec_a), mut(Gr->fin_w), mut(Gr->fin_b), B, H, W, sc, cus, s)))
            return rc;
        if ((rc = hrn_launch_f32_to_planes(fg, dsn, nf * 2, nf, s))) return rc;
    } else if ((rc = hrn_launch_decoder_bwd((const float*)at(tws, L.stack[L.T]), d_sr, Pr->dec_w, Pr->dec_b, Pr->dec_a, Pr->fin_w, dsn,
                                            mut(Gr->dec_w), mut(Gr->dec_b), mut(Gr->dec_a), mut(Gr->fin_w), mut(Gr->fin_b), B, H, W, sc, cus, s)))
        return rc;

    // ---- fusion levels, last to first                                                HRNet.py:113-132
    for (int t = L.T - 1; t >= 0; --t) {
        const int n = L.n_in[t], half = n / 2, pair_last = n - (n & 1) - 1, Mh = B * half;
        const float* st = (const float*)at(tws, L.stack[t]);
        // every G buffer holds B*V*hw*64 floats; Mh <= B*V/2, so one buffer also holds an [Mh][hw][128] tensor
        float* y1 = G[1];                   // d t2               [Mh][hw][128]; dead before ds is written into the same buffer
        float* ds = G[1];                   // gradient of the views entering the level  [B*n][hw][64]
        float* x1 = G[2];                   // df / gC            [Mh][hw][64]
        float* y3 = G[3];                   // d t1 / gA          [Mh][hw][128]
        float* y2 = G[4];                   // gB, later dz       [Mh][hw][128]
        if ((rc = hrn_launch_fuse_df(dsn, alphas, V, pair_last, half, alpha_residual, x1, hw, B, s, dt))) return rc;
        // f = PReLU(convC(t2))
        if ((rc = pre(128, 64, at(tws, L.t2[t]), at(pk, P.fout_w), (const float*)at(pk, P.fout_b), Pr->fuse_out_a, Mh))) return rc;
        if ((rc = hrn_launch_prelu_bwd_bias(x1, (const float*)at(tws, L.f[t]), xpre, Pr->fuse_out_a, x1, (size_t)Mh * hw, 64, mut(Gr->fuse_out_a), mut(Gr->fuse_out_b), sc, s, dt))) return rc;
        if ((rc = conv_wgrad(dt, (const float*)at(tws, L.t2[t]), nullptr, 0, 0, 0, B, x1, Mh, H, W, 128, 64, mut(Gr->fuse_out_w), sc, cus, s))) return rc;
        if ((rc = conv_dgrad(dt, 128, 64, Pr->fuse_out_w, x1, y1, nullptr, Mh, H, W, tws, L, s))) return rc;
        // t2 = z + u, u = PReLU(convB(t1))
        if ((rc = pre(128, 128, at(tws, L.t1[t]), at(pk, P.fres_w[1]), (const float*)at(pk, P.fres_b[1]), Pr->fuse_res_a[1], Mh))) return rc;
        if ((rc = hrn_launch_prelu_bwd_bias(y1, (const float*)at(tws, L.u[t]), xpre, Pr->fuse_res_a[1], y2, (size_t)Mh * hw, 128, mut(Gr->fuse_res_a[1]), mut(Gr->fuse_res_b[1]), sc, s, dt))) return rc;
        if ((rc = conv_wgrad(dt, (const float*)at(tws, L.t1[t]), nullptr, 0, 0, 0, B, y2, Mh, H, W, 128, 128, mut(Gr->fuse_res_w[1]), sc, cus, s))) return rc;
        if ((rc = conv_dgrad(dt, 128, 128, Pr->fuse_res_w[1], y2, y3, nullptr, Mh, H, W, tws, L, s))) return rc;
        // t1 = PReLU(convA(z))
        {
            ConvParams q = conv_base(Mh, H, W);
            q.in_pair = 1; q.stack = st; q.pair_h = half; q.pair_last = pair_last; q.pair_vs = n;
            q.out = xpre; q.wpk = at(pk, P.fres_w[0]); q.bias = (const float*)at(pk, P.fres_b[0]); q.only_if_nonpos = Pr->fuse_res_a[0];
            q.stack_lo = lo_of(dt, (size_t)B * n * hw * 64); q.out_lo = lo_of(dt, (size_t)Mh * hw * 128);
            if ((rc = hrn_launch_conv3x3(dt, 128, 128, q, s))) return rc;
        }
        if ((rc = hrn_launch_prelu_bwd_bias(y3, (const float*)at(tws, L.t1[t]), xpre, Pr->fuse_res_a[0], y3, (size_t)Mh * hw, 128, mut(Gr->fuse_res_a[0]), mut(Gr->fuse_res_b[0]), sc, s, dt))) return rc;
        if ((rc = conv_wgrad(dt, nullptr, st, half, pair_last, n, B, y3, Mh, H, W, 128, 128, mut(Gr->fuse_res_w[0]), sc, cus, s))) return rc;
        if ((rc = conv_dgrad(dt, 128, 128, Pr->fuse_res_w[0], y3, y2, y1, Mh, H, W, tws, L, s))) return rc;     // dz = d t2 + dgradA(gA)
        // dz -> the two views of each pair (+ the alice pass-through)
        if ((rc = hrn_launch_fuse_scatter(dsn, y2, n, half, pair_last, alpha_residual, ds, hw, B, s, dt))) return rc;
        float* tmp = G[0]; G[0] = G[1]; G[1] = tmp;
        dsn = G[0];
    }

    // ---- encoder                                                                     HRNet.py:51-60,62-74
    float* dA = G[1];
    if ((rc = conv_wgrad(dt, (const float*)at(tws, L.a[nl]), nullptr, 0, 0, 0, B, dsn, M, H, W, 64, 64, mut(Gr->enc_final_w), sc, cus, s))) return rc;
    if ((rc = hrn_launch_colsum(dsn, (size_t)M * hw, 64, mut(Gr->enc_final_b), sc, s, dt))) return rc;
    if ((rc = conv_dgrad(dt, 64, 64, Pr->enc_final_w, dsn, dA, nullptr, M, H, W, tws, L, s))) return rc;
    float* e2 = G[2];
    float* e3 = G[3];
    for (int l = nl - 1; l >= 0; --l) {
        // a_{l+1} = a_l + r_l,  r_l = PReLU(conv2(h_l)),  h_l = PReLU(conv1(a_l))
        if ((rc = pre(64, 64, at(tws, L.h[l]), at(pk, P.enc_w[2 * l + 1]), (const float*)at(pk, P.enc_b[2 * l + 1]), Pr->enc_res_a[2 * l + 1], M))) return rc;
        if ((rc = hrn_launch_prelu_bwd_bias(dA, (const float*)at(tws, L.r[l]), xpre, Pr->enc_res_a[2 * l + 1], e2, (size_t)M * hw, 64, mut(Gr->enc_res_a[2 * l + 1]), mut(Gr->enc_res_b[2 * l + 1]), sc, s, dt))) return rc;
        if ((rc = conv_wgrad(dt, (const float*)at(tws, L.h[l]), nullptr, 0, 0, 0, B, e2, M, H, W, 64, 64, mut(Gr->enc_res_w[2 * l + 1]), sc, cus, s))) return rc;
        if ((rc = conv_dgrad(dt, 64, 64, Pr->enc_res_w[2 * l + 1], e2, e3, nullptr, M, H, W, tws, L, s))) return rc;
        if ((rc = pre(64, 64, at(tws, L.a[l]), at(pk, P.enc_w[2 * l]), (const float*)at(pk, P.enc_b[2 * l]), Pr->enc_res_a[2 * l], M))) return rc;
        if ((rc = hrn_launch_prelu_bwd_bias(e3, (const float*)at(tws, L.h[l]), xpre, Pr->enc_res_a[2 * l], e3, (size_t)M * hw, 64, mut(Gr->enc_res_a[2 * l]), mut(Gr->enc_res_b[2 * l]), sc, s, dt))) return rc;
        if ((rc = conv_wgrad(dt, (const float*)at(tws, L.a[l]), nullptr, 0, 0, 0, B, e3, M, H, W, 64, 64, mut(Gr->enc_res_w[2 * l]), sc, cus, s))) return rc;
        if ((rc = conv_dgrad(dt, 64, 64, Pr->enc_res_w[2 * l], e3, e2, dA, M, H, W, tws, L, s))) return rc;       // d a_l = d a_{l+1} + dgrad1(g1)
        float* tmp = dA; dA = e2; e2 = tmp;
    }
    // stem: a_0 = PReLU(conv(cat(view, reference frame)))                               HRNet.py:200-204, :51-53
    if ((rc = hrn_launch_stem_pre(lrs, hw, (const float*)at(tws, L.ref), V, hw, (const float*)at(pk, P.stem_w), (const float*)at(pk, P.stem_b), xpre, M, H, W,
                                  Pr->enc_init_a, s, dt))) return rc;
    if ((rc = hrn_launch_prelu_bwd_bias(dA, (const float*)at(tws, L.a[0]), xpre, Pr->enc_init_a, dA, (size_t)M * hw, 64, mut(Gr->enc_init_a), mut(Gr->enc_init_b), sc, s, dt))) return rc;
    return hrn_launch_stem_wgrad(lrs, hw, (const float*)at(tws, L.ref), V, hw, dA, M, H, W, mut(Gr->enc_init_w), sc, cus, s, dt);
}

}  // extern "C"
