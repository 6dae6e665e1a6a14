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
// fc1 (ShiftNet.py:44,69-72): y[b][j] = sum_k x[b][k] w[j][k], K = 32,768, 1,024 neurons, <= 32 samples per pass: 134 MB of weights read
// once, nothing else of size: an HBM-bound kernel.  Round 2's version loaded the MFMA operand layout straight from global memory - lane r
// = weight row r, 16 bytes per lane: 64 scattered 16-byte requests per instruction - and stayed at 1.8 TB/s whatever its depth.  Here the
// weights (and the activations) go through LDS by DMA in the order they lie in memory: one 1 KB instruction = 256 consecutive k of ONE
// row, 64 rows (32 neurons + 32 samples) per stage, the next stage in flight while this one is multiplied: 64 KB per CU on the wire,
// every request a run of whole lines.  LDS rows are 1,040 bytes apart (16 rows = 64 distinct banks for the operand reads).
// One workgroup = 32 neurons x one K slice of 4,096 (16 stages); wave v multiplies k = 64 v .. 64 v + 63 of a stage on the exact-fp32
// MFMA (v_mfma_f32_32x32x2_f32: lane (r, hh) supplies x[sample r][k + hh], w[neuron r][k + hh]) and writes its own partial slab:
// 8 slices x 4 waves = 32 slabs, summed in fixed order by fc1_finish_kernel.  Workgroup id = slice + 8 x neuron block: the workgroups of
// a slice share an XCD, whose L2 then holds that slice of the activations (512 KB) for all 32 of them.
constexpr int FC_K = 32768;
constexpr int FC_SLICES = 8, FC_KS = FC_K / FC_SLICES;       // K slices, k per slice
constexpr int FC_KST = 256, FC_NST = FC_KS / FC_KST;          // k per stage (1 KB of a row), stages per slice
constexpr int FC_SPLIT = FC_SLICES * 4;                       // partial slabs: one per (slice, wave)
constexpr int FC_PITCH = 1024 + 16;                           // LDS row pitch in bytes
constexpr int FC_STAGE_BYTES = 64 * FC_PITCH;                 // 32 weight rows, then 32 activation rows
constexpr int FC_LDS_BYTES = 2 * FC_STAGE_BYTES;              // double buffer: 133,120 B
typedef __attribute__((address_space(3))) void* fc_lds_ptr;
__global__ __launch_bounds__(256) void fc1_mfma_kernel(const float* __restrict__ xr, const float* __restrict__ w, float* __restrict__ partial, int B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sl = blockIdx.x & (FC_SLICES - 1), nb = blockIdx.x / FC_SLICES;
    // rows of a stage: 0..31 = neurons nb*32 + row, 32..63 = samples row - 32 (beyond the batch: the last valid sample, never stored).
    // Wave v fetches rows v, v + 4, ...: lane l takes bytes 16 l .. 16 l + 15 of the row's 1 KB
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(w + (size_t)nb * 32 * FC_K), 0, (int)(32u * FC_K * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xr, 0, (int)((unsigned)B * FC_K * 4u), 0x00020000);
    auto fetch = [&](int buf, int st) __attribute__((always_inline)) {
        const unsigned kofs = (unsigned)((sl * FC_KS + st * FC_KST) * 4);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = wv + 4 * i;                                   // i < 8: weight rows, i >= 8: activation rows (wave-uniform)
            unsigned char* dst = fc_smem + buf * FC_STAGE_BYTES + row * FC_PITCH;
            if (i < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (fc_lds_ptr)dst, 16, (unsigned)(lane * 16), kofs + (unsigned)row * (FC_K * 4u), 0, 2);
            else {
                const int smp = row - 32 < B ? row - 32 : B - 1;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (fc_lds_ptr)dst, 16, (unsigned)(lane * 16), kofs + (unsigned)smp * (FC_K * 4u), 0, 0);
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    fetch(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int st = 0; st < FC_NST; ++st) {
        const int buf = st & 1;
        if (st + 1 < FC_NST) fetch(buf ^ 1, st + 1);
        const unsigned char* wrow = fc_smem + buf * FC_STAGE_BYTES + r * FC_PITCH + (64 * wv + 4 * hh) * 4;
        const unsigned char* xrow = wrow + 32 * FC_PITCH;
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) {
            const f32x4 wq = *(const f32x4*)(wrow + sb * 32), xq = *(const f32x4*)(xrow + sb * 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xq[e], wq[e], acc, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the next stage has landed (this wave's share; the barrier covers the others')
        __syncthreads();
    }
    // element 4 g + e of lane (r, hh) = sample 8 g + 4 hh + e, neuron r: 32 lanes write 128 contiguous bytes
    float* pp = partial + ((size_t)(sl * 4 + wv) * 32) * 1024 + nb * 32 + r;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) pp[(size_t)(8 * g + 4 * hh + e) * 1024] = acc[4 * g + e];
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
            { const int rc_lds = hrn_allow_lds((const void*)fc1_mfma_kernel, FC_LDS_BYTES); if (rc_lds) return rc_lds; }
            hipLaunchKernelGGL(fc1_mfma_kernel, dim3(32 * FC_SLICES), dim3(256), FC_LDS_BYTES, stream, xr + (size_t)b0 * FC_K, w, partial, nb);
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
