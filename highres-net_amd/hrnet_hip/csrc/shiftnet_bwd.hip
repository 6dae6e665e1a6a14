// ShiftNet training path (fp32): train-mode forward that keeps every layer's tensors, and the backward
// (src/DeepNetworks/ShiftNet.py:16-75; `shifts = register_batch(regis_model, ...)` ... `loss.backward()`, train.py:176-190).
//
// Forward per layer i (ShiftNet.py:16-41): xpre_i = conv_i(ypost_{i-1});  v = BN_train(xpre_i);  ypost_i = [MaxPool2](ReLU(v)).
// Backward per layer, from d ypost_i:
//   d v      = gradient routed to the arg-max of each 2x2 window (first maximum in row-major order, as torch) where v > 0
//   BN       d gamma = sum d v * xhat,  d beta = sum d v,  d xpre = gamma * invstd * (d v - d beta / N - xhat * d gamma / N)
//            (two passes over xpre: reduce, then apply; ReLU / pool masks are recomputed from xpre, nothing extra is stored)
//   conv     bias: column sums; weights: the exact-fp32 MFMA weight-gradient kernel (backward.hip); data: forward conv on W^T
//   stem     2 -> 64 layer: VALU weight gradient on the mean-subtracted input, VALU data gradient back to the two input planes,
//            then the per-plane mean subtraction's own backward (g - mean(g))
// Tail (ShiftNet.py:43-47, :69-74): theta = fc2(ReLU(fc1(dropout(flatten)))) with the reference's (C, H, W) flatten order.
// All reductions are two-stage with a fixed order (deterministic).
#include "../../../include/hrnet_hip.h"
#include "kernels.h"
#include "backward.h"
#include "hrnet_layout.h"
#include "shiftnet_layout.h"

using namespace hrn;

namespace {

constexpr int FCK = 32768;

// ------------------------------------------------------------------------------------------------ BatchNorm (train) helpers
__global__ __launch_bounds__(1024) void bn_save_stats_kernel(const double* __restrict__ partial, int nblk, size_t npix, int C, float eps, float* __restrict__ mean,
                                     float* __restrict__ invstd) {
    const int c = threadIdx.x;
    double s, ss;
    block_pair_sum(partial, nblk, C, s, ss);
    if (c >= C) return;
    const double n = (double)npix, m = s / n;
    double var = ss / n - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean[c] = (float)m;
    invstd[c] = 1.0f / sqrtf((float)var + eps);
}

// d v for the 4 channels c..c+3 of output pixel `op` at its POOL x POOL input positions; returns x values too
template <int POOL>
__device__ __forceinline__ void bn_dv(const float* __restrict__ x, const float* __restrict__ dy, const f32x4 sc, const f32x4 sh, size_t n,
                                      int yo, int xo, int H, int W, int C, int c, size_t op, f32x4 (&xv)[POOL * POOL],
                                      f32x4 (&dv)[POOL * POOL]) {
    const f32x4 g = *(const f32x4*)(dy + op * C + c);
    f32x4 best;
    int arg[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < POOL * POOL; ++k) {
        const int dyy = k / POOL, dxx = k % POOL;
        const size_t ip = (n * H + (size_t)(yo * POOL + dyy)) * W + (xo * POOL + dxx);
        xv[k] = *(const f32x4*)(x + ip * C + c);
        f32x4 v = xv[k] * sc + sh;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float r = fmaxf(v[j], 0.f);
            if (k == 0 || r > best[j]) { best[j] = r; arg[j] = k; }
        }
    }
#pragma unroll
    for (int k = 0; k < POOL * POOL; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) dv[k][j] = (arg[j] == k && best[j] > 0.f) ? g[j] : 0.f;
}

// partial[blk][c] = (sum d v, sum d v * xhat)
template <int POOL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ stats, int N, int H, int W, int C,
                                                            double* __restrict__ partial) {
    __shared__ double red[2][4][256];
    const int c4n = C / 4, lanes = 256 / c4n;
    const int c = (threadIdx.x % c4n) * 4, ln = threadIdx.x / c4n;
    const int Ho = H / POOL, Wo = W / POOL;
    const size_t total = (size_t)N * Ho * Wo;
    const f32x4 mean = *(const f32x4*)(stats + c), istd = *(const f32x4*)(stats + 128 + c);
    const f32x4 sc = *(const f32x4*)(stats + 256 + c), sh = *(const f32x4*)(stats + 384 + c);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    for (size_t op = (size_t)blockIdx.x * lanes + ln; op < total; op += (size_t)gridDim.x * lanes) {
        const int xo = (int)(op % Wo), yo = (int)((op / Wo) % Ho);
        const size_t n = op / ((size_t)Wo * Ho);
        f32x4 xv[POOL * POOL], dv[POOL * POOL];
        bn_dv<POOL>(x, dy, sc, sh, n, yo, xo, H, W, C, c, op, xv, dv);
#pragma unroll
        for (int k = 0; k < POOL * POOL; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s1[j] += (double)dv[k][j];
                s2[j] += (double)dv[k][j] * (double)((xv[k][j] - mean[j]) * istd[j]);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][j][threadIdx.x] = s1[j]; red[1][j][threadIdx.x] = s2[j]; }
    __syncthreads();
    if (ln == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a = 0.0, b = 0.0;
            for (int l = 0; l < lanes; ++l) { a += red[0][j][l * c4n + threadIdx.x]; b += red[1][j][l * c4n + threadIdx.x]; }
            partial[((size_t)blockIdx.x * C + c + j) * 2 + 0] = a;
            partial[((size_t)blockIdx.x * C + c + j) * 2 + 1] = b;
        }
    }
}
// sums[c] = (sum d v, sum d v xhat);  d beta += , d gamma +=
__global__ __launch_bounds__(1024) void bn_bwd_finish_kernel(const double* __restrict__ partial, int nblk, int C, double* __restrict__ sums,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = threadIdx.x;
    double a, b;
    block_pair_sum(partial, nblk, C, a, b);
    if (c >= C) return;
    sums[c * 2] = a; sums[c * 2 + 1] = b;
    dbeta[c] += (float)a;
    dgamma[c] += (float)b;
}
template <int POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const double* __restrict__ sums, float* __restrict__ dx, int N, int H, int W,
                                                           int C) {
    const int c4n = C / 4;
    const int Ho = H / POOL, Wo = W / POOL;
    const size_t total = (size_t)N * Ho * Wo * c4n;
    const float inv_n = 1.0f / (float)((size_t)N * H * W);
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % c4n) * 4;
        const size_t op = idx / c4n;
        const int xo = (int)(op % Wo), yo = (int)((op / Wo) % Ho);
        const size_t n = op / ((size_t)Wo * Ho);
        const f32x4 mean = *(const f32x4*)(stats + c), istd = *(const f32x4*)(stats + 128 + c);
        const f32x4 sc = *(const f32x4*)(stats + 256 + c), sh = *(const f32x4*)(stats + 384 + c);
        const f32x4 gm = *(const f32x4*)(gamma + c);
        f32x4 xv[POOL * POOL], dv[POOL * POOL];
        bn_dv<POOL>(x, dy, sc, sh, n, yo, xo, H, W, C, c, op, xv, dv);
#pragma unroll
        for (int k = 0; k < POOL * POOL; ++k) {
            const int dyy = k / POOL, dxx = k % POOL;
            const size_t ip = (n * H + (size_t)(yo * POOL + dyy)) * W + (xo * POOL + dxx);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xv[k][j] - mean[j]) * istd[j];
                o[j] = gm[j] * istd[j] * (dv[k][j] - (float)sums[(c + j) * 2] * inv_n - xh * (float)sums[(c + j) * 2 + 1] * inv_n);
            }
            *(f32x4*)(dx + ip * C + c) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ stem data gradient
// d in[m][c2][y][x] = sum_co sum_tap g[m][y - ky + 1][x - kx + 1][co] * w[co][c2][ky][kx]      (planes layout [m][2][H][W])
__global__ __launch_bounds__(256) void stem_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ din,
                                                         int M, int H, int W) {
    __shared__ float ws[64 * 18];
    for (int i = threadIdx.x; i < 64 * 18; i += 256) ws[i] = w[i];
    __syncthreads();
    const size_t hw = (size_t)H * W, total = (size_t)M * hw;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t m = idx / hw;
        const int y = (int)((idx - m * hw) / W), x = (int)(idx - m * hw - (size_t)y * W);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int gy = y - ky + 1, gx = x - kx + 1;
                if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                    const f32x4* gp = (const f32x4*)(g + ((m * H + gy) * (size_t)W + gx) * 64);
                    const int tap = ky * 3 + kx;
#pragma unroll 4
                    for (int q = 0; q < 16; ++q) {
                        const f32x4 gv = gp[q];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int co = 4 * q + j;
                            a0 += gv[j] * ws[co * 18 + tap];
                            a1 += gv[j] * ws[co * 18 + 9 + tap];
                        }
                    }
                }
            }
        din[(m * 2 + 0) * hw + (size_t)y * W + x] = a0;
        din[(m * 2 + 1) * hw + (size_t)y * W + x] = a1;
    }
}
// per plane: out = g - mean(g)   (backward of x - mean(x), ShiftNet.py:58); means precomputed
__global__ __launch_bounds__(256) void sub_plane_mean_kernel(const float* __restrict__ g, const float* __restrict__ means, float* __restrict__ out,
                                                             size_t hw, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) out[i] = g[i] - means[i / hw];
}

// ------------------------------------------------------------------------------------------------ fully connected tail
// dy[b][hw*128 + c] = dxr[b][c*256 + hw] * (mask ? 2 * mask : 1)
__global__ __launch_bounds__(256) void fc_from_ref_kernel(const float* __restrict__ dxr, const unsigned char* __restrict__ mask,
                                                          float* __restrict__ dy, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / FCK;
        const int k = (int)(i - b * FCK), hw = k >> 7, c = k & 127;
        const size_t r = b * FCK + (size_t)c * 256 + hw;
        const float v = dxr[r];
        dy[i] = mask ? (mask[r] ? 2.f * v : 0.f) : v;
    }
}
// thread j: dz1[b][j] = (y1[b][j] > 0) * sum_o dtheta[b][o] w2[o][j];  dw2[o][j] += sum_b dtheta[b][o] y1[b][j];  db1[j] += sum_b dz1
__global__ __launch_bounds__(256) void fc2_bwd_kernel(const float* __restrict__ dtheta, const float* __restrict__ y1,
                                                      const float* __restrict__ w2, float* __restrict__ dz1, float* __restrict__ dw2,
                                                      float* __restrict__ db1, int B) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= 1024) return;
    const float w0 = w2[j], w1 = w2[1024 + j];
    float g0 = 0.f, g1 = 0.f, gb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float t0 = dtheta[b * 2], t1 = dtheta[b * 2 + 1], yv = y1[(size_t)b * 1024 + j];
        const float dz = yv > 0.f ? t0 * w0 + t1 * w1 : 0.f;
        dz1[(size_t)b * 1024 + j] = dz;
        g0 += t0 * yv; g1 += t1 * yv; gb += dz;
    }
    dw2[j] += g0;
    dw2[1024 + j] += g1;
    db1[j] += gb;
}
// dw1[j][k] += sum_b dz1[b][j] * xr[b][k]      grid (FCK / 256, 1024 / 64), B <= 32 per launch.  A thread keeps its column of xr (32
// samples) in registers and walks 64 neurons: xr is read 16 times in all (it was once per neuron: 4.3 GB of L2 reads for a 268 MB
// read-modify-write of dw1); dz1[b][j] is the same for the whole workgroup (scalar loads).
constexpr int FC1_BWD_JT = 64;
__global__ __launch_bounds__(256) void fc1_bwd_w_kernel(const float* __restrict__ dz1, const float* __restrict__ xr, float* __restrict__ dw1,
                                                        int B) {
    const int j0 = blockIdx.y * FC1_BWD_JT, k = blockIdx.x * 256 + threadIdx.x;
    float xv[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) xv[b] = b < B ? xr[(size_t)b * FCK + k] : 0.f;
    for (int jj = 0; jj < FC1_BWD_JT; ++jj) {
        const float* dzj = dz1 + j0 + jj;
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < 32; ++b) s += (b < B ? dzj[(size_t)b * 1024] : 0.f) * xv[b];
        float* o = dw1 + (size_t)(j0 + jj) * FCK + k;
        *o += s;
    }
}
// dxr[b][k] = sum_j dz1[b][j] * w1[j][k]: the 134 MB weight matrix once more, now with k - the contiguous axis - as the MFMA's N index.
// One workgroup = 128 k (256 of them); wave v contracts j = 256 v .. 256 v + 255, two j per step on v_mfma_f32_32x32x2_f32: lane (r, hh)
// supplies A = dz1[sample r][j + hh] (from the transposed copy in LDS, pitch 33: no bank conflicts either way) and B = 16 bytes of row
// j + hh, k = k0 + 4 r .. + 3, i.e. four MFMAs per load and every load instruction two runs of 512 contiguous bytes; eight loads in flight
// per wave.  The four waves' sums meet in LDS (fixed order), a lane stores 16 bytes of one sample's row.  B <= 32 per launch.
// (Round 2's form - one k per lane, 32 broadcast LDS reads and 32 FMAs per weight - took 344 us; this one ~50.)
constexpr int FCX_PITCH = 33, FCX_LDS_BYTES = 1024 * FCX_PITCH * 4, FCX_DEPTH = 8;
__global__ __launch_bounds__(256) void fc1_bwd_x_kernel(const float* __restrict__ dz1, const float* __restrict__ w1, float* __restrict__ dxr,
                                                         int B) {
    extern __shared__ __attribute__((aligned(16))) float fcx_smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5, wv = threadIdx.x >> 6;
    const int k0 = blockIdx.x * 128;
    for (int idx = threadIdx.x; idx < 32 * 1024; idx += 256) {
        const int b = idx >> 10, jj = idx & 1023;
        fcx_smem[jj * FCX_PITCH + b] = b < B ? dz1[(size_t)b * 1024 + jj] : 0.f;
    }
    __syncthreads();
    const float* wp = w1 + (size_t)(256 * wv + hh) * FCK + k0 + 4 * r;
    const float* ap = fcx_smem + (256 * wv + hh) * FCX_PITCH + r;
    f32x16 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[e][i] = 0.f;
    f32x4 wq[FCX_DEPTH];
#pragma unroll
    for (int d = 0; d < FCX_DEPTH; ++d) wq[d] = __builtin_nontemporal_load((const f32x4*)(wp + (size_t)(2 * d) * FCK));
    for (int st = 0; st < 128; st += FCX_DEPTH) {
#pragma unroll
        for (int d = 0; d < FCX_DEPTH; ++d) {
            const float a = ap[(2 * (st + d)) * FCX_PITCH];
            const f32x4 wcur = wq[d];
            if (st + FCX_DEPTH + d < 128) wq[d] = __builtin_nontemporal_load((const f32x4*)(wp + (size_t)(2 * (st + FCX_DEPTH + d)) * FCK));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wcur[e], acc[e], 0, 0, 0);
        }
    }
    // element 4 g + e' of lane (r, hh) of acc[e] = sample 8 g + 4 hh + e', k = k0 + 4 r + e.  Waves 1-3 hand theirs over through LDS
    // (the transposed dz1 is no longer needed): slot [wave - 1][element][e][lane]
    __syncthreads();
    if (wv > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) fcx_smem[(((wv - 1) * 16 + i) * 4 + e) * 64 + lane] = acc[e][i];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int b = 8 * (i >> 2) + 4 * hh + (i & 3);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = ((acc[e][i] + fcx_smem[((0 * 16 + i) * 4 + e) * 64 + lane]) + fcx_smem[((1 * 16 + i) * 4 + e) * 64 + lane]) +
                       fcx_smem[((2 * 16 + i) * 4 + e) * 64 + lane];
            if (b < B) *(f32x4*)(dxr + (size_t)b * FCK + k0 + 4 * r) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ workspace
struct SnTrainWs {
    int hin[8], hout[8];
    size_t means, xpre[8], ypost[8], stats[8], y1, partial;
    size_t ga, gb, xr, dxr, dz1, sums, wt, wtp, zero_bias, dxin, gmeans, fc_partial, scratch;
    size_t total;
};
int sn_cus() { return hrn_device_cus(); }
SnTrainWs sn_train_ws(int B) {
    SnTrainWs w;
    memset(&w, 0, sizeof w);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = hrn_align_up(off + bytes, 256); return o; };
    w.means = take((size_t)B * 2 * 4);
    int h = 128;
    size_t big = 0;
    for (int i = 0; i < 8; ++i) {
        w.hin[i] = h;
        w.hout[i] = SN_POOL[i] ? h / 2 : h;
        const size_t a = (size_t)B * h * h * SN_CO[i] * 4, o = (size_t)B * w.hout[i] * w.hout[i] * SN_CO[i] * 4;
        w.xpre[i] = take(a);
        w.ypost[i] = take(o);
        w.stats[i] = take(4 * 128 * 4);
        if (a > big) big = a;
        h = w.hout[i];
    }
    w.y1 = take((size_t)B * 1024 * 4);
    w.partial = take((size_t)SN_PARTIAL_BLOCKS * 128 * 2 * 8);
    w.ga = take(big); w.gb = take(big);
    w.xr = take((size_t)B * FCK * 4); w.dxr = take((size_t)B * FCK * 4); w.dz1 = take((size_t)B * 1024 * 4);
    w.sums = take(128 * 2 * 8);
    w.wt = take((size_t)128 * 128 * 9 * 4); w.wtp = take((size_t)128 * 128 * 9 * 4); w.zero_bias = take(128 * 4);
    w.dxin = take((size_t)B * 2 * 128 * 128 * 4); w.gmeans = take((size_t)B * 2 * 4);
    w.fc_partial = take(hrn_fc1_partial_bytes());
    w.scratch = take(hrn_bwd_scratch_bytes(sn_cus()));
    w.total = off;
    return w;
}
int ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" {

size_t hrn_shiftnet_train_workspace_bytes(int B) { return B > 0 ? sn_train_ws(B).total : 0; }

int hrn_shiftnet_forward_train(const void* packed, const hrn_shiftnet_params* P, const float* x, int B, float momentum,
                               const unsigned char* dropout_mask, float* theta, void* tws, size_t tws_bytes, void* stream) {
    HRN_CHECK(packed && P && x && theta && tws, -2, "hrn_shiftnet_forward_train: null argument");
    HRN_CHECK(B > 0, -2, "hrn_shiftnet_forward_train: empty batch");
    const SnLayout L = sn_layout();
    const SnTrainWs T = sn_train_ws(B);
    HRN_CHECK(tws_bytes >= T.total, -3, "hrn_shiftnet_forward_train: workspace too small (%zu < %zu)", tws_bytes, T.total);
    hipStream_t s = (hipStream_t)stream;
    float* means = (float*)at(tws, T.means);
    double* partial = (double*)at(tws, T.partial);
    const size_t plane = 128 * 128;
    int rc;
    if ((rc = hrn_launch_plane_mean(x, means, B * 2, plane, s))) return rc;                      // ShiftNet.py:58
    for (int i = 0; i < 8; ++i) {
        HRN_CHECK(P->bn_g[i] && P->bn_b[i] && P->bn_rm[i] && P->bn_rv[i], -2, "hrn_shiftnet_forward_train: null BatchNorm tensor %d", i);
        const int h = T.hin[i], C = SN_CO[i];
        float* xp = (float*)at(tws, T.xpre[i]);
        float* yp = (float*)at(tws, T.ypost[i]);
        float* st = (float*)at(tws, T.stats[i]);
        if (i == 0) {
            if ((rc = hrn_launch_stem(HRN_F32, x, 2 * plane, x + plane, 1, 2 * plane, means, (const float*)at(packed, L.conv_w[0]),
                                      (const float*)at(packed, L.conv_b[0]), nullptr, xp, B, h, h, s))) return rc;
        } else {
            ConvParams p = conv_base(B, h, h);
            p.in = at(tws, T.ypost[i - 1]); p.out = xp;
            p.wpk = at(packed, L.conv_w[i]); p.bias = (const float*)at(packed, L.conv_b[i]);
            if ((rc = hrn_launch_conv3x3(HRN_F32, SN_CI[i], C, p, s))) return rc;
        }
        const size_t npix = (size_t)B * h * h;
        if ((rc = hrn_launch_bn_stats(xp, npix, C, P->bn_g[i], P->bn_b[i], 1e-5f, st + 256, st + 384, P->bn_rm[i], P->bn_rv[i], momentum,
                                      partial, SN_PARTIAL_BLOCKS, s))) return rc;
        hipLaunchKernelGGL(bn_save_stats_kernel, dim3(1), dim3(1024), 0, s, (const double*)partial, SN_PARTIAL_BLOCKS, npix, C, 1e-5f, st, st + 128);
        HRN_LAUNCH_CHECK();
        if ((rc = hrn_launch_bn_act_pool(xp, st + 256, st + 384, yp, B, h, h, C, SN_POOL[i], s))) return rc;
    }
    float* y1 = (float*)at(tws, T.y1);
    // fc1's input in the reference's flatten order, dropout folded in: kept in the workspace - the backward's weight gradient reads it
    HRN_CHECK(P->fc1_w, -2, "hrn_shiftnet_forward_train: params->fc1_w is null (fc1.weight is read in place)");
    float* xr = (float*)at(tws, T.xr);
    if ((rc = hrn_launch_fc_to_ref((const float*)at(tws, T.ypost[7]), dropout_mask, xr, B, s))) return rc;
    if ((rc = hrn_launch_fc1(xr, P->fc1_w, (const float*)at(packed, L.fc1_b), y1, B, (float*)at(tws, T.fc_partial), s))) return rc;
    return hrn_launch_fc2(y1, (const float*)at(packed, L.fc2_w), theta, B, s);
}

int hrn_shiftnet_backward(const hrn_shiftnet_params* P, const float* x, int B, const unsigned char* dropout_mask, const float* d_theta,
                          const hrn_shiftnet_params* G, float* d_x, void* tws, size_t tws_bytes, void* stream) {
    HRN_CHECK(P && G && x && d_theta && tws, -2, "hrn_shiftnet_backward: null argument");
    HRN_CHECK(B > 0, -2, "hrn_shiftnet_backward: empty batch");
    const SnTrainWs T = sn_train_ws(B);
    HRN_CHECK(tws_bytes >= T.total, -3, "hrn_shiftnet_backward: workspace too small (%zu < %zu)", tws_bytes, T.total);
    hipStream_t s = (hipStream_t)stream;
    auto mut = [](const float* p) { return const_cast<float*>(p); };
    const int cus = sn_cus();
    void* sc = at(tws, T.scratch);
    double* partial = (double*)at(tws, T.partial);
    double* sums = (double*)at(tws, T.sums);
    float* cur = (float*)at(tws, T.ga);
    float* oth = (float*)at(tws, T.gb);
    float* xr = (float*)at(tws, T.xr);
    float* dxr = (float*)at(tws, T.dxr);
    float* dz1 = (float*)at(tws, T.dz1);
    int rc;
    HRN_HIP(hipMemsetAsync(at(tws, T.zero_bias), 0, 128 * 4, s));
    // ---- tail: theta = fc2(ReLU(fc1(dropout(flatten(y8)))))                                ShiftNet.py:69-74
    hipLaunchKernelGGL(fc2_bwd_kernel, dim3(4), dim3(256), 0, s, d_theta, (const float*)at(tws, T.y1), P->fc2_w, dz1, mut(G->fc2_w), mut(G->fc1_b), B);
    const size_t nflat = (size_t)B * FCK;
    // (xr, the fc1 input in the reference's flatten order with the dropout folded in, was left in the workspace by the forward)
    for (int b0 = 0; b0 < B; b0 += 32)
        hipLaunchKernelGGL(fc1_bwd_w_kernel, dim3(FCK / 256, 1024 / FC1_BWD_JT), dim3(256), 0, s, (const float*)dz1 + (size_t)b0 * 1024,
                           (const float*)xr + (size_t)b0 * FCK, mut(G->fc1_w), B - b0 < 32 ? B - b0 : 32);
    { const int rc_lds = hrn_allow_lds((const void*)fc1_bwd_x_kernel, FCX_LDS_BYTES); if (rc_lds) return rc_lds; }
    for (int b0 = 0; b0 < B; b0 += 32)      // 32 samples are the MFMA's M: larger batches go in groups
        hipLaunchKernelGGL(fc1_bwd_x_kernel, dim3(FCK / 128), dim3(256), FCX_LDS_BYTES, s, (const float*)dz1 + (size_t)b0 * 1024, P->fc1_w,
                           dxr + (size_t)b0 * FCK, B - b0 < 32 ? B - b0 : 32);
    hipLaunchKernelGGL(fc_from_ref_kernel, dim3(ew_grid(nflat)), dim3(256), 0, s, (const float*)dxr, dropout_mask, cur, nflat);
    HRN_LAUNCH_CHECK();
    // ---- layers 8 .. 1                                                                        ShiftNet.py:16-41, :59-67
    for (int i = 7; i >= 0; --i) {
        const int h = T.hin[i], C = SN_CO[i];
        const float* xp = (const float*)at(tws, T.xpre[i]);
        const float* st = (const float*)at(tws, T.stats[i]);
        const size_t npix = (size_t)B * h * h;
        if (SN_POOL[i]) hipLaunchKernelGGL(bn_bwd_reduce_kernel<2>, dim3(SN_PARTIAL_BLOCKS), dim3(256), 0, s, xp, (const float*)cur, st, B, h, h, C, partial);
        else hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(SN_PARTIAL_BLOCKS), dim3(256), 0, s, xp, (const float*)cur, st, B, h, h, C, partial);
        hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(1), dim3(1024), 0, s, (const double*)partial, SN_PARTIAL_BLOCKS, C, sums, mut(G->bn_g[i]), mut(G->bn_b[i]));
        const int eg = ew_grid(npix * C / 4 / (SN_POOL[i] ? 4 : 1));
        if (SN_POOL[i]) hipLaunchKernelGGL(bn_bwd_apply_kernel<2>, dim3(eg), dim3(256), 0, s, xp, (const float*)cur, st, P->bn_g[i], (const double*)sums, oth, B, h, h, C);
        else hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(eg), dim3(256), 0, s, xp, (const float*)cur, st, P->bn_g[i], (const double*)sums, oth, B, h, h, C);
        HRN_LAUNCH_CHECK();
        // oth = d xpre_i
        if ((rc = hrn_launch_colsum(oth, npix, C, mut(G->conv_b[i]), sc, s))) return rc;
        if (i > 0) {
            if ((rc = hrn_launch_conv_wgrad((const float*)at(tws, T.ypost[i - 1]), nullptr, 0, 0, 0, 0, oth, B, h, h, SN_CI[i], C, mut(G->conv_w[i]), sc, cus, s))) return rc;
            if ((rc = hrn_conv_dgrad(SN_CI[i], C, P->conv_w[i], oth, cur, nullptr, B, h, h, (float*)at(tws, T.wt), at(tws, T.wtp),
                                     (const float*)at(tws, T.zero_bias), s))) return rc;
        } else {
            const size_t plane = 128 * 128;
            if ((rc = hrn_launch_stem_wgrad_sub(x, 2 * plane, x + plane, 1, 2 * plane, (const float*)at(tws, T.means), oth, B, h, h, mut(G->conv_w[0]), sc, cus, s))) return rc;
            if (d_x) {
                float* dxin = (float*)at(tws, T.dxin);
                float* gm = (float*)at(tws, T.gmeans);
                hipLaunchKernelGGL(stem_dgrad_kernel, dim3(ew_grid(npix)), dim3(256), 0, s, (const float*)oth, P->conv_w[0], dxin, B, h, h);
                HRN_LAUNCH_CHECK();
                if ((rc = hrn_launch_plane_mean(dxin, gm, B * 2, plane, s))) return rc;
                hipLaunchKernelGGL(sub_plane_mean_kernel, dim3(ew_grid((size_t)B * 2 * plane)), dim3(256), 0, s, (const float*)dxin, (const float*)gm, d_x, plane, (size_t)B * 2 * plane);
                HRN_LAUNCH_CHECK();
            }
        }
    }
    return 0;
}

}  // extern "C"
