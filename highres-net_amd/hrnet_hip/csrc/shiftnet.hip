// ShiftNet tail kernels: BatchNorm statistics / fold, BN+ReLU(+MaxPool2), fc1(+dropout mask)+ReLU, fc2.
//   /root/reference/src/DeepNetworks/ShiftNet.py:16-47 (layers), :49-75 (forward)
// The 3x3 convolutions run on conv3x3.hip (f32 MFMA path), the 2->64 stem and the per-plane mean on stem.hip.
// Activations are NHWC f32; fc1 reads its weight in place, in the reference's flatten order (the 4 MB input is re-ordered instead).
#include "kernels.h"

namespace {

// ---- BatchNorm2d batch statistics (train mode): deterministic two-stage reduction in fp64
// stage 1: partial[blk][c] = (sum, sumsq) over a contiguous slice of pixels
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, size_t npix, int C, double* __restrict__ partial) {
    __shared__ double red[2][256];
    const int groups = 256 / C;                 // C = 64 -> 4 pixel groups, C = 128 -> 2
    const int c = threadIdx.x % C, g = threadIdx.x / C;
    const size_t per_blk = (npix + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per_blk;
    const size_t p1 = p0 + per_blk < npix ? p0 + per_blk : npix;
    double s = 0.0, ss = 0.0;
    for (size_t p = p0 + g; p < p1; p += groups) {
        const double v = (double)x[p * C + c];
        s += v; ss += v * v;
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = ss;
    __syncthreads();
    if (g == 0) {
        for (int k = 1; k < groups; ++k) { s += red[0][k * C + c]; ss += red[1][k * C + c]; }
        partial[((size_t)blockIdx.x * C + c) * 2 + 0] = s;
        partial[((size_t)blockIdx.x * C + c) * 2 + 1] = ss;
    }
}
// stage 2: mean / biased var -> scale, shift; running stats (momentum, unbiased var) updated in place when given
__global__ __launch_bounds__(1024) void bn_finish_kernel(const double* __restrict__ partial, int nblk, size_t npix, int C,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                 float* __restrict__ scale, float* __restrict__ shift,
                                 float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
    const int c = threadIdx.x;
    double s, ss;
    block_pair_sum(partial, nblk, C, s, ss);
    if (c >= C) return;
    const double n = (double)npix;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float sc = gamma[c] / sqrtf((float)var + eps);
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (running_mean) {
        const double unbiased = npix > 1 ? var * n / (n - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}
// eval mode: fold running statistics.  With conv_bias the convolution's own bias goes into the shift as well: y = conv_nobias * scale +
// shift is then the whole of BatchNorm(conv(x)) and runs as the conv kernel's epilogue (ConvParams::scale / bias / relu)
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                               const float* __restrict__ rv, float eps, const float* __restrict__ conv_bias, float* __restrict__ scale,
                               float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] + ((conv_bias ? conv_bias[c] : 0.f) - rm[c]) * sc;
}

// ---- y = max(0, x*scale + shift), optionally followed by MaxPool2d(2).  One thread = 4 channels of one output pixel.
template <int POOL>
__global__ __launch_bounds__(256) void bn_act_pool_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ out,
                                                          int N, int H, int W, int C) {
    const int Ho = H / POOL, Wo = W / POOL, c4n = C / 4;
    const size_t total = (size_t)N * Ho * Wo * c4n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const size_t op = idx / c4n;
        const int xo = (int)(op % Wo);
        const int yo = (int)((op / Wo) % Ho);
        const size_t n = op / ((size_t)Wo * Ho);
        const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 sc = scale ? *(const f32x4*)(scale + c) : one, sh = shift ? *(const f32x4*)(shift + c) : zero;     // null: pool (+ ReLU) only
        f32x4 best;
#pragma unroll
        for (int dy = 0; dy < POOL; ++dy)
#pragma unroll
            for (int dx = 0; dx < POOL; ++dx) {
                const size_t ip = (n * H + (size_t)(yo * POOL + dy)) * W + (xo * POOL + dx);
                f32x4 v = *(const f32x4*)(x + ip * C + c) * sc + sh;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                if (dy == 0 && dx == 0) best = v;
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) best[j] = fmaxf(best[j], v[j]);
                }
            }
        *(f32x4*)(out + op * C + c) = best;
    }
}

// ---- fc1: y[b][j] = relu(bias[j] + sum_k xr[b][k] * w[j][k]),  K = 32768, J = 1024            (ShiftNet.py:44, :69-72)
// A 32 x 1024 x 32768 GEMM whose time is ONE read of the 134 MB weight matrix.  Both operands are in the REFERENCE's flatten order
// k = c*256 + hw: the weights are read in place (no packed copy, nothing to re-pack after an optimiser step) and the 4 MB input is
// brought into that order - dropout folded in - by fc_to_ref_kernel (the backward wants exactly that tensor too).
// Exact-fp32 MFMA 32x32x2: D[m = sample][n = neuron].  K is split into FC_SPLIT = 32 slices of 1,024; one wave = one (neuron block of
// 32, slice): 1,024 waves, four per CU.  Per 32-deep super-step a lane (r, hh) loads 64 contiguous bytes of its weight row r (so every
// row is consumed in whole 128-byte lines) and of its sample row r: k = ks + 16 hh + t feeds MFMA t of the step on both sides - the
// order in which a product's k is visited is free.  Weights are read once: non-temporal; xr (L2 resident) by plain loads.  The
// slices' partial sums land in `partial` [FC_SPLIT][32][1024] and fc1_finish_kernel adds them in slice order: bit-reproducible.
#ifndef FC_SPLIT_N
#define FC_SPLIT_N 64
#endif
#ifndef FC_NBW
#define FC_NBW 2          // neuron blocks of 32 per wave (they share the wave's xr fragments)
#endif
#ifndef FC_DEPTH
#define FC_DEPTH 2        // super-steps of loads in flight
#endif
constexpr int FC_K = 32768, FC_SPLIT = FC_SPLIT_N, FC_KS = FC_K / FC_SPLIT;
__global__ __launch_bounds__(256) void fc1_mfma_kernel(const float* __restrict__ xr, const float* __restrict__ w, float* __restrict__ partial, int B) {
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int NBG = 32 / FC_NBW;                                     // neuron groups
    const int nb = wid % NBG, sl = wid / NBG;
    const float* wp = w + (size_t)(nb * 32 * FC_NBW + r) * FC_K + sl * FC_KS + 16 * hh;
    const float* xp = xr + (size_t)(r < B ? r : B - 1) * FC_K + sl * FC_KS + 16 * hh;      // rows beyond the batch: any valid row, never stored
    f32x16 acc[FC_NBW];
#pragma unroll
    for (int q = 0; q < FC_NBW; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    f32x4 wv[FC_DEPTH][FC_NBW][4], xv[FC_DEPTH][4];
    auto fetch = [&](int buf, int step) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int q = 0; q < FC_NBW; ++q) wv[buf][q][i] = __builtin_nontemporal_load((const f32x4*)(wp + (size_t)q * 32 * FC_K + step * 32 + 4 * i));
            xv[buf][i] = *(const f32x4*)(xp + step * 32 + 4 * i);
        }
    };
    auto multiply = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < FC_NBW; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[buf][i][e], wv[buf][q][i][e], acc[q], 0, 0, 0);
    };
    constexpr int NS = FC_KS / 32;                                       // super-steps of the slice
    static_assert(NS % FC_DEPTH == 0, "slice length");
#pragma unroll
    for (int d = 0; d < FC_DEPTH; ++d) fetch(d, d);
    for (int st = 0; st < NS; st += FC_DEPTH) {
#pragma unroll
        for (int d = 0; d < FC_DEPTH; ++d) {
            multiply(d);
            if (st + FC_DEPTH + d < NS) fetch(d, st + FC_DEPTH + d);
        }
    }
    // element 4 g + e of lane (r, hh) = sample 8 g + 4 hh + e, neuron r: 32 lanes write 128 contiguous bytes
#pragma unroll
    for (int q = 0; q < FC_NBW; ++q) {
        float* pp = partial + ((size_t)sl * 32) * 1024 + (nb * FC_NBW + q) * 32 + r;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[(size_t)(8 * g + 4 * hh + e) * 1024] = acc[q][4 * g + e];
    }
}
__global__ __launch_bounds__(256) void fc1_finish_kernel(const float* __restrict__ partial, const float* __restrict__ bias, float* __restrict__ y, int B) {
    const int idx = blockIdx.x * 256 + threadIdx.x;                      // b * 1024 + j
    if (idx >= B * 1024) return;
    float v = bias[idx & 1023];
#pragma unroll 8
    for (int s = 0; s < FC_SPLIT; ++s) v += partial[(size_t)s * 32 * 1024 + idx];
    y[idx] = fmaxf(v, 0.f);
}

// xr[b][c*256 + hw] = y[b][hw*128 + c] * (mask ? 2 * mask[b][c*256 + hw] : 1): the fc1 input in the reference's flatten order, the
// train-mode dropout (p = 0.5, kept activations x 2) folded in.  Through an LDS tile so that both sides move whole lines.
__global__ __launch_bounds__(256) void fc_to_ref_kernel(const float* __restrict__ y, const unsigned char* __restrict__ mask,
                                                        float* __restrict__ xr) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;               // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = y[(size_t)b * FC_K + (size_t)(hw0 + ty + 8 * i) * 128 + c0 + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t o = (size_t)b * FC_K + (size_t)(c0 + ty + 8 * i) * 256 + hw0 + tx;
        const float v = tile[tx][ty + 8 * i];
        xr[o] = mask ? (mask[o] ? 2.f * v : 0.f) : v;
    }
}

// fc2: theta[b][o] = sum_j y[b][j] * w2[o][j]  (no bias; ShiftNet.py:46)
__global__ __launch_bounds__(256) void fc2_kernel(const float* __restrict__ y, const float* __restrict__ w2, float* __restrict__ theta) {
    __shared__ float red[2][4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f;
    for (int j = threadIdx.x; j < 1024; j += 256) {
        const float v = y[(size_t)b * 1024 + j];
        a0 += v * w2[j];
        a1 += v * w2[1024 + j];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); }
    if (lane == 0) { red[0][wave] = a0; red[1][wave] = a1; }
    __syncthreads();
    if (threadIdx.x < 2) theta[b * 2 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

}  // namespace

int hrn_launch_bn_stats(const float* x, size_t npix, int C, const float* gamma, const float* beta, float eps,
                        float* scale, float* shift, float* running_mean, float* running_var, float momentum,
                        double* partial, int partial_blocks, hipStream_t stream) {
    HRN_CHECK(C == 64 || C == 128, -2, "bn_stats: unsupported channel count %d", C);
    HRN_CHECK(partial_blocks > 0, -2, "bn_stats: no partial buffer");
    HrnProfScope prof("bn_stats", 0.0, (double)npix * C * 4, stream);
    hipLaunchKernelGGL(bn_partial_kernel, dim3(partial_blocks), dim3(256), 0, stream, x, npix, C, partial);
    HRN_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finish_kernel, dim3(1), dim3(1024), 0, stream, partial, partial_blocks, npix, C, gamma, beta, eps,
                       scale, shift, running_mean, running_var, momentum);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                       const float* conv_bias, float* scale, float* shift, int C, hipStream_t stream) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3(1), dim3(128), 0, stream, gamma, beta, rm, rv, eps, conv_bias, scale, shift, C);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_bn_act_pool(const float* x, const float* scale, const float* shift, float* out, int N, int H, int W, int C,
                           int pool, hipStream_t stream) {
    const int p = pool ? 2 : 1;
    HRN_CHECK(!pool || (H % 2 == 0 && W % 2 == 0), -2, "maxpool2 needs even H, W");
    const size_t total = (size_t)N * (H / p) * (W / p) * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    HrnProfScope prof("bn_relu_pool", 0.0, (double)N * H * W * C * 4 * (1.0 + 1.0 / (p * p)), stream);
    if (pool) hipLaunchKernelGGL(bn_act_pool_kernel<2>, dim3(blocks), dim3(256), 0, stream, x, scale, shift, out, N, H, W, C);
    else hipLaunchKernelGGL(bn_act_pool_kernel<1>, dim3(blocks), dim3(256), 0, stream, x, scale, shift, out, N, H, W, C);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_fc_to_ref(const float* y, const unsigned char* mask, float* xr, int B, hipStream_t stream) {
    HrnProfScope prof("fc_to_ref", 0.0, (double)B * FC_K * 8, stream);
    hipLaunchKernelGGL(fc_to_ref_kernel, dim3(256 / 32, 128 / 32, B), dim3(256), 0, stream, y, mask, xr);
    HRN_LAUNCH_CHECK();
    return 0;
}

size_t hrn_fc1_partial_bytes(void) { return (size_t)FC_SPLIT * 32 * 1024 * 4; }

// any batch size: groups of 32 samples (each group streams the weights once)
int hrn_launch_fc1(const float* xr, const float* w, const float* b, float* y, int B, float* partial, hipStream_t stream) {
    for (int b0 = 0; b0 < B; b0 += 32) {
        const int nb = B - b0 < 32 ? B - b0 : 32;
        {
            HrnProfScope prof("fc1", 2.0 * nb * 1024 * 32768, 1024.0 * 32768 * 4 + (double)nb * 32768 * 4, stream);
            hipLaunchKernelGGL(fc1_mfma_kernel, dim3(32 / FC_NBW * FC_SPLIT / 4), dim3(256), 0, stream, xr + (size_t)b0 * FC_K, w, partial, nb);
        }
        hipLaunchKernelGGL(fc1_finish_kernel, dim3((nb * 1024 + 255) / 256), dim3(256), 0, stream, (const float*)partial, b, y + (size_t)b0 * 1024, nb);
        HRN_LAUNCH_CHECK();
    }
    return 0;
}

int hrn_launch_fc2(const float* y, const float* w, float* theta, int B, hipStream_t stream) {
    hipLaunchKernelGGL(fc2_kernel, dim3(B), dim3(256), 0, stream, y, w, theta);
    HRN_LAUNCH_CHECK();
    return 0;
}
