// ShiftNet tail kernels: BatchNorm statistics / fold, BN+ReLU(+MaxPool2), fc1(+dropout mask)+ReLU, fc2.
//   /root/reference/src/DeepNetworks/ShiftNet.py:16-47 (layers), :49-75 (forward)
// The 3x3 convolutions run on conv3x3.hip (f32 MFMA path), the 2->64 stem and the per-plane mean on stem.hip.
// Activations are NHWC f32; fc1's weight is re-ordered once to the NHWC flatten order at pack time.
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
__global__ void bn_finish_kernel(const double* __restrict__ partial, int nblk, size_t npix, int C,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                 float* __restrict__ scale, float* __restrict__ shift,
                                 float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < nblk; ++b) { s += partial[((size_t)b * C + c) * 2]; ss += partial[((size_t)b * C + c) * 2 + 1]; }
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
// eval mode: fold running statistics
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                               const float* __restrict__ rv, float eps, float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
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
        const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
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

// ---- fc1: y[b][j] = relu(bias[j] + sum_k x[b][k] * w[j][k]),  K = 32768 (NHWC flatten: k = hw*128 + c), J = 1024.
// One workgroup per FC_NJ = 4 output neurons streams their four 128 KB weight rows once per batch tile of FC_NB = 16 samples; x
// (4 MB at B = 32) stays L2 resident and is read 1024 / FC_NJ times (it was once per NEURON: 4 GB of L2 reads, 2.0 ms).
// mask (optional, train-mode dropout p=0.5): uint8 [B][32768] in the REFERENCE's flatten order c*256 + hw; x is scaled by 2*mask.
// Summation order: thread-strided partial dot products, a fixed shuffle tree, then the four waves in order (bit-reproducible).
constexpr int FC_K = 32768, FC_NJ = 4, FC_NB = 16;
__global__ __launch_bounds__(256) void fc1_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                  const unsigned char* __restrict__ mask, float* __restrict__ y, int B) {
    __shared__ float red[FC_NJ][FC_NB][4];
    const int j0 = blockIdx.x * FC_NJ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b0 = 0; b0 < B; b0 += FC_NB) {
        float acc[FC_NJ][FC_NB];
#pragma unroll
        for (int n = 0; n < FC_NJ; ++n)
#pragma unroll
            for (int i = 0; i < FC_NB; ++i) acc[n][i] = 0.f;
        for (int k4 = threadIdx.x; k4 < FC_K / 4; k4 += 256) {
            f32x4 wv[FC_NJ];
#pragma unroll
            for (int n = 0; n < FC_NJ; ++n) wv[n] = *(const f32x4*)(w + (size_t)(j0 + n) * FC_K + (size_t)k4 * 4);
#pragma unroll
            for (int i = 0; i < FC_NB; ++i) {
                const int b = b0 + i;
                if (b < B) {
                    f32x4 xv = *(const f32x4*)(x + (size_t)b * FC_K + (size_t)k4 * 4);
                    if (mask) {
                        const int k = k4 * 4, hw = k >> 7, c = k & 127;
                        const unsigned char* mk = mask + (size_t)b * FC_K + (size_t)c * 256 + hw;
#pragma unroll
                        for (int e = 0; e < 4; ++e) xv[e] = mk[e * 256] ? xv[e] * 2.f : 0.f;
                    }
#pragma unroll
                    for (int n = 0; n < FC_NJ; ++n) acc[n][i] += xv[0] * wv[n][0] + xv[1] * wv[n][1] + xv[2] * wv[n][2] + xv[3] * wv[n][3];
                }
            }
        }
#pragma unroll
        for (int n = 0; n < FC_NJ; ++n)
#pragma unroll
            for (int i = 0; i < FC_NB; ++i) {
                float v = acc[n][i];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                if (lane == 0) red[n][i][wave] = v;
            }
        __syncthreads();
        if (threadIdx.x < FC_NJ * FC_NB) {
            const int n = threadIdx.x / FC_NB, i = threadIdx.x % FC_NB;
            if (b0 + i < B) {
                const float v = red[n][i][0] + red[n][i][1] + red[n][i][2] + red[n][i][3] + bias[j0 + n];
                y[(size_t)(b0 + i) * 1024 + j0 + n] = fmaxf(v, 0.f);
            }
        }
        __syncthreads();
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

// fc1.weight [1024][c*256 + hw] -> [1024][hw*128 + c]
__global__ void fc1_pack_kernel(const float* __restrict__ w, float* __restrict__ out) {
    const size_t total = (size_t)1024 * FC_K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t j = idx / FC_K;
        const int k = (int)(idx - j * FC_K);
        const int hw = k >> 7, c = k & 127;
        out[idx] = w[j * FC_K + (size_t)c * 256 + hw];
    }
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
    hipLaunchKernelGGL(bn_finish_kernel, dim3(1), dim3(128), 0, stream, partial, partial_blocks, npix, C, gamma, beta, eps,
                       scale, shift, running_mean, running_var, momentum);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                       const float* /*conv_bias*/, float* scale, float* shift, int C, hipStream_t stream) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3(1), dim3(128), 0, stream, gamma, beta, rm, rv, eps, scale, shift, C);
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

int hrn_launch_fc1(const float* x, const float* w, const float* b, const unsigned char* mask, float* y, int B, hipStream_t stream) {
    HrnProfScope prof("fc1", 2.0 * B * 1024 * 32768, 1024.0 * 32768 * 4 + (double)B * 32768 * 4, stream);
    hipLaunchKernelGGL(fc1_kernel, dim3(1024 / FC_NJ), dim3(256), 0, stream, x, w, b, mask, y, B);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_fc2(const float* y, const float* w, float* theta, int B, hipStream_t stream) {
    hipLaunchKernelGGL(fc2_kernel, dim3(B), dim3(256), 0, stream, y, w, theta);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_fc1_pack(const float* w, float* packed, hipStream_t stream) {
    hipLaunchKernelGGL(fc1_pack_kernel, dim3(4096), dim3(256), 0, stream, w, packed);
    HRN_LAUNCH_CHECK();
    return 0;
}
