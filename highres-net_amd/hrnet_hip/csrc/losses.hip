// Score / loss reductions that sit right behind the hot path:
//   masked_cmse  <- get_loss(srs, hrs, hr_maps, metric)      /root/reference/src/train.py:66-87  (+ get_crop_mask :90-106)
//   shift_cpsnr  <- shift_cPSNR(sr, hr, hr_map, border_w=3)   /root/reference/src/Evaluator.py:52-73 (cPSNR :11-43)
// Both are pure HBM-bound reductions: every pixel is read once (f1: 12 B/px; f2: the 49 shifted windows hit L2).
// The brightness-corrected MSE needs the bias b = sum(m (hr - sr)) / n before the squared error; with d = sr - hr
//   sum m (d + b)^2 = S2 - S1^2 / S0,   S0 = sum m, S1 = sum m d, S2 = sum m d^2
// so ONE pass with three fp64 accumulators replaces the reference's two passes (and is exact to fp64 rounding).
// (Evaluator.cPSNR squares ((diff - bias) * map): identical for the binary status maps it is defined on; the kernel
// takes m^2 for that term when `square_mask` is set, to follow it literally.)
#include "kernels.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

// one workgroup per sample.  srs/hrs/maps [B][S][S]; crop: border width whose mask is forced to 0 (get_crop_mask).
// metric: 0 masked_MSE (mean over all pixels of (m sr - m hr)^2), 1 cMSE, 2 -10 log10(cMSE)
__global__ __launch_bounds__(256) void masked_cmse_kernel(const float* __restrict__ srs, const float* __restrict__ hrs,
                                                          const float* __restrict__ maps, int S, int crop, int metric,
                                                          float* __restrict__ out) {
    __shared__ double red[4];
    const size_t n = (size_t)S * S;
    const float* sr = srs + (size_t)blockIdx.x * n;
    const float* hr = hrs + (size_t)blockIdx.x * n;
    const float* mp = maps + (size_t)blockIdx.x * n;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = (int)(i / S), x = (int)(i - (size_t)y * S);
        float m = mp[i];
        if (y < crop || y >= S - crop || x < crop || x >= S - crop) m = 0.f;
        const double d = (double)sr[i] - (double)hr[i];
        if (metric == 0) { const double e = (double)m * d; s2 += e * e; }
        else { s0 += m; s1 += (double)m * d; s2 += (double)m * d * d; }
    }
    s0 = block_sum(s0, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        if (metric == 0) { out[blockIdx.x] = (float)(s2 / (double)n); return; }
        const double cmse = (s2 - s1 * s1 / s0) / s0;
        out[blockIdx.x] = metric == 1 ? (float)cmse : (float)(-10.0 * log10(cmse));
    }
}

// grid (shifts = (2b+1)^2, B).  sr is compared as its centre crop [b:b+size]^2 against hr / map at [u:u+size, v:v+size].
__global__ __launch_bounds__(256) void shift_cpsnr_kernel(const float* __restrict__ srs, const float* __restrict__ hrs,
                                                          const float* __restrict__ maps, int S, int border, int clip,
                                                          double* __restrict__ scores) {
    __shared__ double red[4];
    const int nb = 2 * border + 1;
    const int u = blockIdx.x / nb, v = blockIdx.x - u * nb;      // row / column offset (get_patch(img, x=u, y=v))
    const int size = S - 2 * border;
    const size_t n = (size_t)S * S;
    const float* sr = srs + (size_t)blockIdx.y * n;
    const float* hr = hrs + (size_t)blockIdx.y * n;
    const float* mp = maps + (size_t)blockIdx.y * n;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < size * size; i += blockDim.x) {
        const int y = i / size, x = i - y * size;
        float s = sr[(size_t)(y + border) * S + (x + border)];
        if (clip) s = fminf(fmaxf(s, 0.f), 1.f);                  // np.clip(sr, 0, 1) at the call sites (predict.py:43, train.py:212)
        const size_t j = (size_t)(y + u) * S + (x + v);
        const double m = (double)mp[j];
        const double d = (double)hr[j] - (double)s;              // Evaluator.py:35 diff = hr - sr
        s0 += m; s1 += m * d; s2 += m * m * d * d;               // :37 squares (diff - bias) * map
    }
    s0 = block_sum(s0, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red);
    // sum ((d - b) m)^2 with b = S1 / S0 needs sum m^2 d and sum m^2 too for non-binary maps; status maps are binary
    // (m^2 == m), for which it equals S2 - S1^2 / S0
    if (threadIdx.x == 0) {
        const double cmse = (s2 - s1 * s1 / s0) / s0;
        scores[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = -10.0 * log10(cmse);
    }
}

__global__ void shift_max_kernel(const double* __restrict__ scores, int nshift, float* __restrict__ out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double best = scores[(size_t)b * nshift];
    for (int i = 1; i < nshift; ++i) best = fmax(best, scores[(size_t)b * nshift + i]);
    out[b] = (float)best;
}

// ---- the registered-loss tail of a training step, differentiable (train.py:78-87 and :183-187): forward in two stages (per-slice
// fp64 partial sums, fixed-order finish: bit-reproducible), backward one elementwise pass.  With d = sr - hr, S0 = sum m,
// S1 = sum m d, S2 = sum m d^2:  n = S0, b = -S1 / S0 (the brightness bias, DETACHED in the reference: train.py:83),
// cMSE = (S2 - S1^2 / S0) / S0, and d cMSE / d sr_i = 2 m_i (d_i + b) / n.
constexpr int LOSS_SPLIT = 16;      // slices per sample: B x 16 workgroups stream the three images once

__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ srs, const float* __restrict__ hrs,
                                                           const float* __restrict__ maps, int S, int crop, double* __restrict__ partial) {
    __shared__ double red[4];
    const size_t n = (size_t)S * S;
    const size_t per = (n + LOSS_SPLIT - 1) / LOSS_SPLIT;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    const float* sr = srs + (size_t)blockIdx.y * n;
    const float* hr = hrs + (size_t)blockIdx.y * n;
    const float* mp = maps + (size_t)blockIdx.y * n;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int y = (int)(i / S), x = (int)(i - (size_t)y * S);
        float m = mp[i];
        if (y < crop || y >= S - crop || x < crop || x >= S - crop) m = 0.f;
        const double d = (double)sr[i] - (double)hr[i];
        s0 += m; s1 += (double)m * d; s2 += (double)m * d * d;
    }
    s0 = block_sum(s0, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        double* o = partial + ((size_t)blockIdx.y * LOSS_SPLIT + blockIdx.x) * 3;
        o[0] = s0; o[1] = s1; o[2] = s2;
    }
}

__global__ void loss_finish_kernel(const double* __restrict__ partial, int B, int metric, float* __restrict__ out, double* __restrict__ stats) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < LOSS_SPLIT; ++k) {
        const double* o = partial + ((size_t)b * LOSS_SPLIT + k) * 3;
        s0 += o[0]; s1 += o[1]; s2 += o[2];
    }
    const double cmse = (s2 - s1 * s1 / s0) / s0;
    stats[4 * b + 0] = s0; stats[4 * b + 1] = -s1 / s0; stats[4 * b + 2] = cmse; stats[4 * b + 3] = 0.0;
    out[b] = metric == 1 ? (float)cmse : (float)(-10.0 * log10(cmse));
}

__global__ __launch_bounds__(256) void loss_backward_kernel(const float* __restrict__ srs, const float* __restrict__ hrs,
                                                            const float* __restrict__ maps, const double* __restrict__ stats,
                                                            const float* __restrict__ d_out, int S, int crop, int metric,
                                                            float* __restrict__ d_srs) {
    const size_t n = (size_t)S * S;
    const int b = blockIdx.y;
    const double cnt = stats[4 * b + 0], bias = stats[4 * b + 1], cmse = stats[4 * b + 2];
    // d out / d cMSE: cMSE itself -> 1;  -10 log10(cMSE) -> -10 / (ln 10 cMSE)
    const double dm = metric == 1 ? 1.0 : -10.0 / (2.302585092994046 * cmse);
    const float coef = (float)((double)d_out[b] * dm * 2.0 / cnt);
    const float fb = (float)bias;
    const size_t base = (size_t)b * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / S), x = (int)(i - (size_t)y * S);
        float m = maps[base + i];
        if (y < crop || y >= S - crop || x < crop || x >= S - crop) m = 0.f;
        d_srs[base + i] = coef * m * (srs[base + i] - hrs[base + i] + fb);
    }
}

}  // namespace

size_t hrn_loss_train_workspace_bytes_impl(int B) { return (size_t)B * LOSS_SPLIT * 3 * sizeof(double); }

int hrn_launch_loss_train(const float* srs, const float* hrs, const float* maps, int B, int S, int crop, int metric, float* out,
                          double* stats, double* partial, hipStream_t stream) {
    HrnProfScope prof("registered_loss_fwd", 0.0, 12.0 * B * S * S, stream);
    hipLaunchKernelGGL(loss_partial_kernel, dim3(LOSS_SPLIT, B), dim3(256), 0, stream, srs, hrs, maps, S, crop, partial);
    HRN_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, (const double*)partial, B, metric, out, stats);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_loss_backward(const float* srs, const float* hrs, const float* maps, const double* stats, const float* d_out, int B,
                             int S, int crop, int metric, float* d_srs, hipStream_t stream) {
    HrnProfScope prof("registered_loss_bwd", 0.0, 16.0 * B * S * S, stream);
    int gx = (int)(((size_t)S * S + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(loss_backward_kernel, dim3(gx, B), dim3(256), 0, stream, srs, hrs, maps, stats, d_out, S, crop, metric, d_srs);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_masked_cmse(const float* srs, const float* hrs, const float* maps, int B, int S, int crop, int metric, float* out,
                           hipStream_t stream) {
    HrnProfScope prof("masked_cmse", 0.0, 12.0 * B * S * S, stream);
    hipLaunchKernelGGL(masked_cmse_kernel, dim3(B), dim3(256), 0, stream, srs, hrs, maps, S, crop, metric, out);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_shift_cpsnr(const float* srs, const float* hrs, const float* maps, int B, int S, int border, int clip,
                           double* scores, float* out, hipStream_t stream) {
    const int nshift = (2 * border + 1) * (2 * border + 1);
    HrnProfScope prof("shift_cpsnr", 0.0, 12.0 * B * S * S, stream);
    hipLaunchKernelGGL(shift_cpsnr_kernel, dim3(nshift, B), dim3(256), 0, stream, srs, hrs, maps, S, border, clip, scores);
    HRN_LAUNCH_CHECK();
    hipLaunchKernelGGL(shift_max_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, scores, nshift, out, B);
    HRN_LAUNCH_CHECK();
    return 0;
}
