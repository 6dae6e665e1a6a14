// Backward building blocks of the HRNet training path (fp32, NHWC): SURVEY.md section 8f row f3, `loss.backward()` through
// HRNet in src/train.py:190.  Everything here is deterministic: reductions go through per-workgroup partial sums in a
// workspace and a fixed-order finish kernel - no float atomics.
//
//   prelu_bwd      g = dy * (y > 0 ? 1 : a), da += sum dy * min(x, 0)          (nn.PReLU, HRNet.py:19,21,53,97,151)
//                  from the POST-activation y the forward stored: x < 0 <=> y < 0 and min(x, 0) = y / a, valid for a > 0
//                  (the reference initialises a = 0.25; a <= 0 is rejected on the Python side)
//   colsum         db[c] += sum over pixels of g[pixel][c]                        (conv bias gradient)
//   conv wgrad     dW[co][ci][ky][kx] += sum_pixels g[p][co] * x[p + (ky-1, kx-1)][ci]   (nn.Conv2d 3x3 pad 1 weight gradient)
//                  exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) with K = pixels: A = g^T, B = shifted x, both read as single
//                  floats from LDS tiles; one wave owns a 32 x 32 (cout, cin) block for all nine taps (144 accumulators)
//                  and keeps it across every tile of its persistent workgroup
//   stem wgrad     the same for the 2 -> 64 stem (fp32 planes as input, VALU)
//   dgrad          needs no kernel: dx = conv3x3(g, W^T with taps flipped) runs on the forward kernel after
//                  hrn_launch_dgrad_weights() has produced the transposed OIHW tensor
#include "kernels.h"
#include "backward.h"

#ifndef WG_ABL
#define WG_ABL 0        // timing-only ablations of conv_wgrad_kernel (results WRONG): 1 no prefetch of the next tile | 2 no LDS staging | 4 no operand reads
#endif

// the elementwise passes read every input byte exactly once: non-temporal (they then leave the L2 to the kernels around them)
#define BWD_LD(p) __builtin_nontemporal_load(p)
#define BWD_ST(p, v) __builtin_nontemporal_store(v, p)

namespace {

constexpr int RED_BLOCKS = 512;         // partial sums per reduction

// ------------------------------------------------------------------------------------------------ scalar reductions
// out[0] += sum of partial[0..n): one 256-thread block, fixed summation tree (deterministic)
__global__ __launch_bounds__(256) void scalar_finish_kernel(const double* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += (float)red[0];
}

// ------------------------------------------------------------------------------------------------ column sums (bias grads)
template <int C, bool X3>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ g, size_t rows, double* __restrict__ partial) {
    // thread -> 4 channels c4*4.., row phase tid / (C/4); 16-byte loads, four rows in flight per thread; fixed order
    constexpr int C4 = C / 4, RP = 256 / C4;
    const int c4 = threadIdx.x % C4, rp = threadIdx.x / C4;
    const size_t lo = rows * C * 2;                             // (X3: the lo plane of the [rows][C] tensor)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const size_t stride = (size_t)gridDim.x * RP;
    size_t r = (size_t)blockIdx.x * RP + rp;
    for (; r + 3 * stride < rows; r += 4 * stride) {
        const f32x4 v0 = act_ld4<X3>(g, lo, r * C4 + c4), v1 = act_ld4<X3>(g, lo, (r + stride) * C4 + c4);
        const f32x4 v2 = act_ld4<X3>(g, lo, (r + 2 * stride) * C4 + c4), v3 = act_ld4<X3>(g, lo, (r + 3 * stride) * C4 + c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += ((double)v0[j] + (double)v1[j]) + ((double)v2[j] + (double)v3[j]);
    }
    for (; r < rows; r += stride) {
        const f32x4 v = act_ld4<X3>(g, lo, r * C4 + c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += (double)v[j];
    }
    __shared__ double red[4][256];
#pragma unroll
    for (int j = 0; j < 4; ++j) red[j][threadIdx.x] = acc[j];
    __syncthreads();
    if (rp == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < RP; ++k) s += red[j][k * C4 + c4];
            partial[(size_t)blockIdx.x * C + c4 * 4 + j] = s;
        }
    }
}
// PReLU backward and the bias gradient of the convolution in front of it in ONE pass over dy / y: g = dy * PReLU'(x),
// colpart[blk][c] = this block's column sums of g, slopepart[blk] = its share of the slope gradient.  Thread mapping and summation
// order of colsum_kernel (fixed: bit-reproducible); g may alias dy.  With a positive slope the sign of x is the sign of the stored
// post-activation y and x = y / slope where it is negative; otherwise (decided here, on the device: no host round trip) the
// pre-activation itself is read from xpre, which the caller has recomputed under the same condition.
template <int C, bool X3>
__global__ __launch_bounds__(256) void prelu_bwd_bias_kernel(const void* __restrict__ dy, const void* __restrict__ y,
                                                             const void* __restrict__ xpre, const float* __restrict__ slope,
                                                             void* __restrict__ g, size_t rows, double* __restrict__ colpart,
                                                             double* __restrict__ slopepart) {
    constexpr int C4 = C / 4, RP = 256 / C4;
    const int c4 = threadIdx.x % C4, rp = threadIdx.x / C4;
    const size_t lo = rows * C * 2;                             // (X3: the lo plane of each [rows][C] tensor)
    const float a = slope[0];
    const bool from_y = a > 0.f;
    const void* __restrict__ src = from_y ? y : xpre;
    const float inv_a = from_y ? 1.f / a : 1.f;
    double acc[4] = {0.0, 0.0, 0.0, 0.0}, sacc = 0.0;
    const size_t stride = (size_t)gridDim.x * RP;
    auto one = [&](size_t r) __attribute__((always_inline)) {
        const size_t o4 = r * C4 + c4;
        const f32x4 d = act_ld4<X3>(dy, lo, o4), v = act_ld4<X3>(src, lo, o4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool pos = v[j] > 0.f;
            o[j] = pos ? d[j] : a * d[j];
            acc[j] += (double)o[j];
            if (!pos) sacc += (double)d[j] * ((double)v[j] * (double)inv_a);      // signed terms that largely cancel: fp64
        }
        if constexpr (X3) act_st4<true>(g, lo, o4, o);
        else *((f32x4*)g + o4) = o;
    };
    size_t r = (size_t)blockIdx.x * RP + rp;
    for (; r + stride < rows; r += 2 * stride) { one(r); one(r + stride); }
    for (; r < rows; r += stride) one(r);
    __shared__ double red[4][256];
    __shared__ double sred[256];
#pragma unroll
    for (int j = 0; j < 4; ++j) red[j][threadIdx.x] = acc[j];
    sred[threadIdx.x] = sacc;
    __syncthreads();
    if (rp == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double t = 0.0;
            for (int k = 0; k < RP; ++k) t += red[j][k * C4 + c4];
            colpart[(size_t)blockIdx.x * C + c4 * 4 + j] = t;
        }
    }
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sred[threadIdx.x] += sred[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) slopepart[blockIdx.x] = sred[0];
}

// out[c] += sum over blocks of partial[b][c]: one block per 32 channels, 32 block-phases per channel (the loop is a chain of loads: its
// length, not the bytes, is what a finish costs), fixed order
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const double* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
    __shared__ double red[32][32];
    const int cl = threadIdx.x & 31, ph = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double s = 0.0;
#pragma unroll 4
    for (int b = ph; b < nblk; b += 32) s += partial[(size_t)b * C + c];
    red[ph][cl] = s;
    __syncthreads();
    if (ph == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][cl];
        out[c] += (float)t;
    }
}

// ------------------------------------------------------------------------------------------------ dgrad weights
// w [co][ci][3][3] -> wt [ci][co][3][3] with taps flipped: wt[ci][co][ky][kx] = w[co][ci][2-ky][2-kx]
__global__ void dgrad_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int cin, int cout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cin * cout * 9;
    if (idx >= total) return;
    const int tap = idx % 9, co = (idx / 9) % cout, ci = idx / (9 * cout);
    wt[idx] = w[((size_t)co * cin + ci) * 9 + (8 - tap)];
}

// ------------------------------------------------------------------------------------------------ conv weight gradient
// One persistent workgroup of 256 threads per CU walks 8 x 32-pixel tiles.  Per tile, LDS holds the g tile [256 px][64 co]
// and the x halo tile [10 x 34 px][64 ci] for one (cout chunk, cin chunk) pair of 64 x 64; wave w owns the 32 x 32 block
// (cb, ib) = (w >> 1, w & 1) for all 9 taps.  MFMA 32x32x2: k = two consecutive pixels of the tile.
constexpr int WG_TH = 8, WG_TW = 32, WG_HW = WG_TW + 2, WG_HH = WG_TH + 2;
constexpr int WG_XP = 64 + 4;           // floats per halo pixel in LDS (+4: the hh = 1 lanes land on other banks)
constexpr int WG_GP = 64 + 4;
constexpr int WG_LDS = (WG_HH * WG_HW * WG_XP + WG_TH * WG_TW * WG_GP) * 4;     // 92,480 + 69,632 = 162,112 B

struct WgradParams {
    const float* x;         // plain input [M][H][W][CIN] (in_pair == 0)
    const float* stack;     // pair gather: views [B][pair_vs][H][W][64]
    int in_pair, pair_h, pair_last, pair_vs;
    const float* g;         // [M][H][W][COUT]
    float* partial;         // [gridDim.x][9][64][64] for this (cout chunk, cin chunk)
    int M, H, W, cin, cout, co_chunk, ci_chunk;
};

__global__ __launch_bounds__(256, 1) void conv_wgrad_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xs = (float*)smem;                                   // [340][WG_XP]
    float* gs = xs + WG_HH * WG_HW * WG_XP;                     // [256][WG_GP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int cb = wave >> 1, ib = wave & 1;
    const int H = p.H, W = p.W;
    const size_t hw = (size_t)H * W;
    const int tiles_x = (W + WG_TW - 1) / WG_TW, tiles_y = (H + WG_TH - 1) / WG_TH;
    const long tiles = (long)tiles_x * tiles_y, total = tiles * p.M;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // The next tile's x halo (340 px x 16 float4) and g tile (256 px x 16 float4) travel HBM -> registers while the current tile
    // is multiplied (one wave per SIMD: 512 registers, 144 of them accumulators), and registers -> LDS between two barriers.
    // Thread -> (pixel column tid / 16 of a 16-pixel segment, 16-byte part tid % 16); a load `it` is a (row, segment) pair that is the
    // same for every thread, so its row test and row address are scalar and a load costs one add: x halo rows 0..9 x segments
    // {0, 1} = halo columns 0..31 (20 loads), the two leftover columns 32, 33 of all ten rows in two more (thread -> its own row);
    // g rows 0..7 x segments {0, 1} (16 loads).
    constexpr int NXR = 2 * WG_HH + 2, NGR = 2 * WG_TH;        // 22, 16 float4 per thread
    f32x4 xreg[NXR], greg[NGR];
    const int pcol = tid >> 4, part = tid & 15;
    // leftover columns: element e = tid + 256 k (k = 0, 1) < 320 -> (row e / 32, column 32 + (e % 32) / 16, part e % 16)
    const int lrow[2] = {tid >> 5, 8 + (tid >> 5)}, lcol = 32 + ((tid >> 4) & 1);
    auto issue = [&](long tl) __attribute__((always_inline)) {
        const int m = (int)(tl / tiles);
        const int t = (int)(tl - (long)m * tiles);
        const int ty = t / tiles_x, y0 = ty * WG_TH, x0 = (t - ty * tiles_x) * WG_TW;
        // source of the cin chunk: plain tensor, or view i / partner view of the pair gather (64 channels each)
        const float* xb;
        int xpitch;
        if (p.in_pair) {
            const int b = m / p.pair_h, i = m - b * p.pair_h;
            const int v = p.ci_chunk == 0 ? i : p.pair_last - i;
            xb = p.stack + ((size_t)b * p.pair_vs + v) * hw * 64;
            xpitch = 64;
        } else {
            xb = p.x + (size_t)m * hw * p.cin + p.ci_chunk * 64;
            xpitch = p.cin;
        }
        const float* gb = p.g + (size_t)m * hw * p.cout + p.co_chunk * 64;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        // per-thread column tests and offsets of the two segments (and of the leftover column), once per tile
        const int gx0 = x0 - 1 + pcol, gx1 = gx0 + 16, gxl = x0 - 1 + lcol;
        const bool okx0 = (unsigned)gx0 < (unsigned)W, okx1 = (unsigned)gx1 < (unsigned)W, okxl = (unsigned)gxl < (unsigned)W;
        const int xo0 = gx0 * xpitch + part * 4, xo1 = gx1 * xpitch + part * 4, xol = gxl * xpitch + part * 4;
#pragma unroll
        for (int row = 0; row < WG_HH; ++row) {
            const int gy = y0 + row - 1;                        // uniform
            const bool oky = (unsigned)gy < (unsigned)H;
            const float* rb = xb + (size_t)(oky ? gy : 0) * W * xpitch;
            xreg[2 * row] = (oky && okx0) ? *(const f32x4*)(rb + xo0) : zero;
            xreg[2 * row + 1] = (oky && okx1) ? *(const f32x4*)(rb + xo1) : zero;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int gy = y0 + lrow[k] - 1;
            const bool ok = (k == 0 || tid < 64) && (unsigned)gy < (unsigned)H && okxl;
            xreg[2 * WG_HH + k] = ok ? *(const f32x4*)(xb + (size_t)gy * W * xpitch + xol) : zero;
        }
        // g: zero outside the image (those pixels contribute nothing)
        const int hx0 = x0 + pcol, hx1 = hx0 + 16;
        const int go0 = hx0 * p.cout + part * 4, go1 = hx1 * p.cout + part * 4;
#pragma unroll
        for (int row = 0; row < WG_TH; ++row) {
            const int gy = y0 + row;
            const bool oky = gy < H;
            const float* rb = gb + (size_t)(oky ? gy : 0) * W * p.cout;
            greg[2 * row] = (oky && hx0 < W) ? *(const f32x4*)(rb + go0) : zero;
            greg[2 * row + 1] = (oky && hx1 < W) ? *(const f32x4*)(rb + go1) : zero;
        }
    };
    if ((long)blockIdx.x < total) issue(blockIdx.x);
    for (long tl = blockIdx.x; tl < total; tl += gridDim.x) {
        __syncthreads();                                        // previous tile's MFMA reads are done
        if (!(WG_ABL & 2) || tl == blockIdx.x) {
#pragma unroll
            for (int row = 0; row < WG_HH; ++row) {
                *(f32x4*)(xs + (row * WG_HW + pcol) * WG_XP + part * 4) = xreg[2 * row];
                *(f32x4*)(xs + (row * WG_HW + 16 + pcol) * WG_XP + part * 4) = xreg[2 * row + 1];
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (k == 0 || tid < 64) *(f32x4*)(xs + (lrow[k] * WG_HW + lcol) * WG_XP + part * 4) = xreg[2 * WG_HH + k];
#pragma unroll
            for (int row = 0; row < WG_TH; ++row) {
                *(f32x4*)(gs + (row * WG_TW + pcol) * WG_GP + part * 4) = greg[2 * row];
                *(f32x4*)(gs + (row * WG_TW + 16 + pcol) * WG_GP + part * 4) = greg[2 * row + 1];
            }
        }
        __syncthreads();
        if (!(WG_ABL & 1)) if (tl + gridDim.x < total) issue(tl + gridDim.x);      // in flight during this tile's MFMAs
        // 128 k-steps of two pixels: A[co = r][k = hh] = g[pixel 2s + hh][cb*32 + r], B[k = hh][ci = r] = x[pixel + tap][ib*32 + r]
        const float* ga = gs + hh * WG_GP + cb * 32 + r;
        const float* xa = xs + hh * WG_XP + ib * 32 + r;
        // operands of k-step s+1 are read while the nine MFMAs of k-step s run (the compiler otherwise waits for each
        // k-step's ten ds_reads right in front of its MFMAs; sched_barrier pins the order) and land in the OTHER of two operand
        // sets: no register copies between k-steps
        float a0, b0[9], a1 = 0.f, b1[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto fetch = [&](int s, float& a, float (&b)[9]) __attribute__((always_inline)) {
            if (WG_ABL & 4) return;
            const int pp = 2 * s;                               // even pixel of the pair; both pixels share the tile row
            const int row = pp >> 5, col = pp & 31;
            a = ga[pp * WG_GP];
            const float* xr = xa + (row * WG_HW + col) * WG_XP;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) b[ky * 3 + kx] = xr[(ky * WG_HW + kx) * WG_XP];
        };
        auto multiply = [&](float a, const float (&b)[9]) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
        };
        if (WG_ABL & 4) { a0 = a1 = ga[0]; for (int t = 0; t < 9; ++t) b0[t] = b1[t] = xa[t * WG_XP]; }
        fetch(0, a0, b0);
        for (int s = 0; s < 128; s += 2) {
            fetch(s + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            multiply(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < 128) fetch(s + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            multiply(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // partial[blk][tap][co 64][ci 64]; accumulator element (g4, j) of lane (r = ci column, hh) is row co = 8*g4 + 4*hh + j
    float* out = p.partial + (size_t)blockIdx.x * 9 * 4096;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = cb * 32 + 8 * g4 + 4 * hh + j;
                out[(size_t)t * 4096 + co * 64 + ib * 32 + r] = acc[t][4 * g4 + j];
            }
}

// dW[co][ci][tap] (OIHW, full cin/cout) += sum over workgroups of partial[blk][tap][co_l][ci_l]
__global__ __launch_bounds__(1024) void wgrad_finish_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ dw, int cin, int co_chunk,
                                    int ci_chunk) {
    // block = 64 consecutive elements x 16 workgroup phases (more loads in flight than one thread per element), fixed order
    __shared__ double red[16][64];
    const int el = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + el;                       // over 9 * 64 * 64
    double s = 0.0;
#pragma unroll 4
    for (int b = ph; b < nblk; b += 16) s += (double)partial[(size_t)b * 9 * 4096 + idx];
    red[ph][el] = s;
    __syncthreads();
    if (ph != 0) return;
    s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][el];
    const int tap = idx / 4096, co_l = (idx >> 6) & 63, ci_l = idx & 63;
    const int co = co_chunk * 64 + co_l, ci = ci_chunk * 64 + ci_l;
    dw[((size_t)co * cin + ci) * 9 + tap] += (float)s;
}

// ------------------------------------------------------------------------------------------------ stem weight gradient
// dW[co][c2][tap] += sum g[m][p][co] * in_c2[m][p + tap], in_0 = view m, in_1 = reference frame of sample m / rep1
template <bool X3>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ in0, size_t stride0, const float* __restrict__ in1,
                                                         int rep1, size_t stride1, const float* __restrict__ sub,
                                                         const void* __restrict__ g, int M, int H, int W, float* __restrict__ partial) {
    __shared__ float tile[2][WG_HH][WG_HW];
    const int tid = threadIdx.x, co = tid & 63, q = tid >> 6;
    const int tiles_x = (W + WG_TW - 1) / WG_TW, tiles_y = (H + WG_TH - 1) / WG_TH;
    const long tiles = (long)tiles_x * tiles_y, total = tiles * M;
    const size_t hw = (size_t)H * W;
    float acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = 0.f;
    for (long tl = blockIdx.x; tl < total; tl += gridDim.x) {
        const int m = (int)(tl / tiles);
        const int t = (int)(tl - (long)m * tiles);
        const int ty = t / tiles_x, y0 = ty * WG_TH, x0 = (t - ty * tiles_x) * WG_TW;
        const float* p0 = in0 + (size_t)m * stride0;
        const float* p1 = in1 + (size_t)(m / rep1) * stride1;
        __syncthreads();
        for (int i = tid; i < 2 * WG_HH * WG_HW; i += 256) {
            const int c = i / (WG_HH * WG_HW), pix = i - c * WG_HH * WG_HW;
            const int py = pix / WG_HW, px = pix - py * WG_HW;
            const int gy = y0 + py - 1, gx = x0 + px - 1;
            float v = 0.f;
            // `sub` (ShiftNet): the forward subtracts the plane mean before the zero padding, ShiftNet.py:58
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = (c == 0 ? p0 : p1)[(size_t)gy * W + gx] - (sub ? sub[m * 2 + c] : 0.f);
            tile[c][py][px] = v;
        }
        __syncthreads();
        const size_t gb = (size_t)m * hw * 64 + co, glo = (size_t)M * hw * 64 * 2;
        // eight of the thread's 64 pixels at a time: their loads of g are in flight together (the loop is latency-bound otherwise);
        // pixels outside the image contribute zero
        for (int p8 = 0; p8 < WG_TH * WG_TW / 4; p8 += 8) {
            float gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pix = q + 4 * (p8 + u), gy = y0 + (pix >> 5), gx = x0 + (pix & 31);
                gv[u] = (gy < H && gx < W) ? act_ld1<X3>(g, glo, gb + ((size_t)gy * W + gx) * 64) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pix = q + 4 * (p8 + u), py = pix >> 5, px = pix & 31;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc[c * 9 + ky * 3 + kx] += gv[u] * tile[c][py + ky][px + kx];
            }
        }
    }
    // partial[blk][q][co][18]
    float* out = partial + ((size_t)blockIdx.x * 4 + q) * 64 * 18 + co * 18;
#pragma unroll
    for (int k = 0; k < 18; ++k) out[k] = acc[k];
}
// dw[idx] += sum over rows of partial[row][idx], idx over 64 * 18 (dw layout [co][c2][3][3] = co*18 + c*9 + tap): 16 outputs x 16 row
// phases per workgroup, fixed order
__global__ __launch_bounds__(1024) void stem_wgrad_finish_kernel(const float* __restrict__ partial, int nrows, float* __restrict__ dw) {
    __shared__ double red[64][16];
    const int o = threadIdx.x & 15, ph = threadIdx.x >> 4, idx = blockIdx.x * 16 + o;
    double s = 0.0;
#pragma unroll 4
    for (int b = ph; b < nrows; b += 64) s += (double)partial[(size_t)b * 64 * 18 + idx];
    red[ph][o] = s;
    __syncthreads();
    if (ph == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 64; ++k) t += red[k][o];
        dw[idx] += (float)t;
    }
}

// ------------------------------------------------------------------------------------------------ elementwise helpers
template <bool X3>
__global__ __launch_bounds__(256) void add_kernel(const void* __restrict__ a, const void* __restrict__ b, void* __restrict__ o, size_t n4) {
    const size_t lo = n4 * 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        act_st4<X3>(o, lo, i, act_ld4<X3>(a, lo, i) + act_ld4<X3>(b, lo, i));
}

// fusion level, forward: s'[b][i] = s[b][i] + alpha[b][partner(i)] * f[b][i]   (or s' = f without the alpha residual)
template <bool X3>
__global__ __launch_bounds__(256) void fuse_update_kernel(const void* __restrict__ stack, int n_in, const void* __restrict__ f,
                                                          const float* __restrict__ alphas, int alpha_vs, int pair_last, int half,
                                                          int alpha_residual, void* __restrict__ out, size_t img4, int B) {
    const size_t total = (size_t)B * half * img4;
    const size_t lo_s = (size_t)B * n_in * img4 * 8, lo_f = total * 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t img = i / img4, e = i - img * img4;
        const int b = (int)(img / half), v = (int)(img - (size_t)b * half);
        const f32x4 fv = act_ld4<X3>(f, lo_f, i);
        if (alpha_residual) {
            const float al = alphas[(size_t)b * alpha_vs + (pair_last - v)];
            act_st4<X3>(out, lo_f, i, act_ld4<X3>(stack, lo_s, ((size_t)b * n_in + v) * img4 + e) + al * fv);
        } else {
            act_st4<X3>(out, lo_f, i, fv);
        }
    }
}
// fusion level, backward: df[b][i] = alpha_partner * ds'[b][i]  (or ds')
template <bool X3>
__global__ __launch_bounds__(256) void fuse_df_kernel(const void* __restrict__ dsn, const float* __restrict__ alphas, int alpha_vs,
                                                      int pair_last, int half, int alpha_residual, void* __restrict__ df, size_t img4,
                                                      int B) {
    const size_t total = (size_t)B * half * img4, lo = total * 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t img = i / img4;
        const int b = (int)(img / half), v = (int)(img - (size_t)b * half);
        const float al = alpha_residual ? alphas[(size_t)b * alpha_vs + (pair_last - v)] : 1.f;
        act_st4<X3>(df, lo, i, al * act_ld4<X3>(dsn, lo, i));
    }
}
// fusion level, backward: gradient of the level's input views from ds' (alice pass-through, alpha residual only) and
// dz [B*half][HW][128] (channels 0..63 -> view i, 64..127 -> view pair_last - i); views that took no part get zero
template <bool X3>
__global__ __launch_bounds__(256) void fuse_scatter_kernel(const void* __restrict__ dsn, const void* __restrict__ dz, int n_in, int half,
                                                           int pair_last, int alpha_residual, void* __restrict__ ds, size_t hw, int B) {
    const size_t img4 = hw * 16;                                // float4 per 64-channel image
    const size_t total = (size_t)B * n_in * img4;
    const size_t lo_n = (size_t)B * half * img4 * 8, lo_z = (size_t)B * half * hw * 32 * 8, lo_s = total * 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t img = i / img4, e = i - img * img4;
        const int b = (int)(img / n_in), v = (int)(img - (size_t)b * n_in);
        const size_t pix = e >> 4, part = e & 15;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (v < half) {
            if (alpha_residual) o = act_ld4<X3>(dsn, lo_n, ((size_t)b * half + v) * img4 + e);
            o += act_ld4<X3>(dz, lo_z, (((size_t)b * half + v) * hw + pix) * 32 + part);
        } else if (v <= pair_last && pair_last - v < half) {
            o = act_ld4<X3>(dz, lo_z, (((size_t)b * half + (pair_last - v)) * hw + pix) * 32 + 16 + part);
        }
        act_st4<X3>(ds, lo_s, i, o);
    }
}

int red_grid(size_t n4) {
    size_t g = (n4 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

size_t hrn_bwd_scratch_bytes(int num_cus) {
    // wgrad partial slabs (one (cout chunk, cin chunk) pair at a time; the bf16x3 kernel runs two workgroups per CU) + reduction partials
    return (size_t)2 * num_cus * 9 * 4096 * 4 + (size_t)RED_BLOCKS * (128 + 1) * 8 + 4096;
}

int hrn_launch_prelu_bwd_bias(const float* dy, const float* y, const float* xpre, const float* slope, float* g, size_t rows, int C,
                              float* dslope, float* db, void* scratch, hipStream_t s, int dt) {
    HRN_CHECK(C == 64 || C == 128, -2, "prelu_bwd_bias: C must be 64 or 128 (got %d)", C);
    double* colpart = (double*)scratch;
    const int blocks = RED_BLOCKS;
    double* slopepart = colpart + (size_t)blocks * 128;
    const bool x3 = dt == HRN_BF16X3;
#define HRN_PB(C_, X_) hipLaunchKernelGGL((prelu_bwd_bias_kernel<C_, X_>), dim3(blocks), dim3(256), 0, s, (const void*)dy, (const void*)y, (const void*)xpre, slope, (void*)g, rows, colpart, slopepart)
    if (C == 64) { if (x3) HRN_PB(64, true); else HRN_PB(64, false); }
    else { if (x3) HRN_PB(128, true); else HRN_PB(128, false); }
#undef HRN_PB
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(C / 32), dim3(1024), 0, s, colpart, blocks, C, db);
    hipLaunchKernelGGL(scalar_finish_kernel, dim3(1), dim3(256), 0, s, slopepart, blocks, dslope);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_colsum(const float* g, size_t rows, int C, float* db, void* scratch, hipStream_t s, int dt) {
    HRN_CHECK(C == 64 || C == 128, -2, "colsum: C must be 64 or 128 (got %d)", C);
    double* partial = (double*)scratch;
    const int blocks = RED_BLOCKS;
    const bool x3 = dt == HRN_BF16X3;
#define HRN_CS(C_, X_) hipLaunchKernelGGL((colsum_kernel<C_, X_>), dim3(blocks), dim3(256), 0, s, (const void*)g, rows, partial)
    if (C == 64) { if (x3) HRN_CS(64, true); else HRN_CS(64, false); }
    else { if (x3) HRN_CS(128, true); else HRN_CS(128, false); }
#undef HRN_CS
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(C / 32), dim3(1024), 0, s, partial, blocks, C, db);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_dgrad_weights(const float* w, float* wt, int cin, int cout, hipStream_t s) {
    const int total = cin * cout * 9;
    hipLaunchKernelGGL(dgrad_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, wt, cin, cout);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_wgrad_finish(const float* partial, int nblk, float* dw, int cin, int co_chunk, int ci_chunk, hipStream_t s) {
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(9 * 4096 / 64), dim3(1024), 0, s, partial, nblk, dw, cin, co_chunk, ci_chunk);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_conv_wgrad(const float* x, const float* stack, int in_pair, int pair_h, int pair_last, int pair_vs, const float* g,
                          int M, int H, int W, int cin, int cout, float* dw, void* scratch, int num_cus, hipStream_t s) {
    HRN_CHECK((cin == 64 || cin == 128) && (cout == 64 || cout == 128), -2, "conv_wgrad: unsupported %d -> %d", cin, cout);
    HRN_CHECK(!in_pair || cin == 128, -2, "conv_wgrad: the pair gather has 128 input channels");
    { const int rc_lds = hrn_allow_lds((const void*)conv_wgrad_kernel, WG_LDS); if (rc_lds) return rc_lds; }
    static_assert(WG_LDS <= 160 * 1024, "LDS budget");
    const long tiles = (long)((W + WG_TW - 1) / WG_TW) * ((H + WG_TH - 1) / WG_TH) * M;
    int grid = num_cus;
    if (tiles < grid) grid = (int)tiles;
    WgradParams p;
    p.x = x; p.stack = stack; p.in_pair = in_pair; p.pair_h = pair_h; p.pair_last = pair_last; p.pair_vs = pair_vs;
    p.g = g; p.partial = (float*)scratch; p.M = M; p.H = H; p.W = W; p.cin = cin; p.cout = cout;
    for (int cc = 0; cc < cout / 64; ++cc)
        for (int ic = 0; ic < cin / 64; ++ic) {
            p.co_chunk = cc; p.ci_chunk = ic;
            hipLaunchKernelGGL(conv_wgrad_kernel, dim3(grid), dim3(256), WG_LDS, s, p);
            hipLaunchKernelGGL(wgrad_finish_kernel, dim3(9 * 4096 / 64), dim3(1024), 0, s, (const float*)scratch, grid, dw, cin, cc, ic);
        }
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_stem_wgrad(const float* in0, size_t stride0, const float* in1, int rep1, size_t stride1, const float* g, int M, int H,
                          int W, float* dw, void* scratch, int num_cus, hipStream_t s, int dt) {
    return hrn_launch_stem_wgrad_sub(in0, stride0, in1, rep1, stride1, nullptr, g, M, H, W, dw, scratch, num_cus, s, dt);
}

int hrn_launch_stem_wgrad_sub(const float* in0, size_t stride0, const float* in1, int rep1, size_t stride1, const float* sub,
                              const float* g, int M, int H, int W, float* dw, void* scratch, int num_cus, hipStream_t s, int dt) {
    const long tiles = (long)((W + WG_TW - 1) / WG_TW) * ((H + WG_TH - 1) / WG_TH) * M;
    int grid = 4 * num_cus;                 // four workgroups per CU: the kernel waits on memory, not on arithmetic
    if (tiles < grid) grid = (int)tiles;
    if (dt == HRN_BF16X3) hipLaunchKernelGGL(stem_wgrad_kernel<true>, dim3(grid), dim3(256), 0, s, in0, stride0, in1, rep1, stride1, sub, (const void*)g, M, H, W, (float*)scratch);
    else hipLaunchKernelGGL(stem_wgrad_kernel<false>, dim3(grid), dim3(256), 0, s, in0, stride0, in1, rep1, stride1, sub, (const void*)g, M, H, W, (float*)scratch);
    hipLaunchKernelGGL(stem_wgrad_finish_kernel, dim3(64 * 18 / 16), dim3(1024), 0, s, (const float*)scratch, grid * 4, dw);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_add(const float* a, const float* b, float* o, size_t n, hipStream_t s, int dt) {
    HRN_CHECK(n % 4 == 0, -2, "add: element count %zu not a multiple of 4", n);
    if (dt == HRN_BF16X3) hipLaunchKernelGGL(add_kernel<true>, dim3(red_grid(n / 4)), dim3(256), 0, s, (const void*)a, (const void*)b, (void*)o, n / 4);
    else hipLaunchKernelGGL(add_kernel<false>, dim3(red_grid(n / 4)), dim3(256), 0, s, (const void*)a, (const void*)b, (void*)o, n / 4);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_fuse_update(const float* stack, int n_in, const float* f, const float* alphas, int alpha_vs, int pair_last, int half,
                           int alpha_residual, float* out, size_t hw, int B, hipStream_t s, int dt) {
    const size_t img4 = hw * 16;
    if (dt == HRN_BF16X3) hipLaunchKernelGGL(fuse_update_kernel<true>, dim3(red_grid((size_t)B * half * img4)), dim3(256), 0, s, (const void*)stack, n_in, (const void*)f, alphas, alpha_vs,
                                             pair_last, half, alpha_residual, (void*)out, img4, B);
    else hipLaunchKernelGGL(fuse_update_kernel<false>, dim3(red_grid((size_t)B * half * img4)), dim3(256), 0, s, (const void*)stack, n_in, (const void*)f, alphas, alpha_vs,
                            pair_last, half, alpha_residual, (void*)out, img4, B);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_fuse_df(const float* dsn, const float* alphas, int alpha_vs, int pair_last, int half, int alpha_residual, float* df,
                       size_t hw, int B, hipStream_t s, int dt) {
    const size_t img4 = hw * 16;
    if (dt == HRN_BF16X3) hipLaunchKernelGGL(fuse_df_kernel<true>, dim3(red_grid((size_t)B * half * img4)), dim3(256), 0, s, (const void*)dsn, alphas, alpha_vs, pair_last, half,
                                             alpha_residual, (void*)df, img4, B);
    else hipLaunchKernelGGL(fuse_df_kernel<false>, dim3(red_grid((size_t)B * half * img4)), dim3(256), 0, s, (const void*)dsn, alphas, alpha_vs, pair_last, half,
                            alpha_residual, (void*)df, img4, B);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_fuse_scatter(const float* dsn, const float* dz, int n_in, int half, int pair_last, int alpha_residual, float* ds,
                            size_t hw, int B, hipStream_t s, int dt) {
    if (dt == HRN_BF16X3) hipLaunchKernelGGL(fuse_scatter_kernel<true>, dim3(red_grid((size_t)B * n_in * hw * 16)), dim3(256), 0, s, (const void*)dsn, (const void*)dz, n_in, half, pair_last,
                                             alpha_residual, (void*)ds, hw, B);
    else hipLaunchKernelGGL(fuse_scatter_kernel<false>, dim3(red_grid((size_t)B * n_in * hw * 16)), dim3(256), 0, s, (const void*)dsn, (const void*)dz, n_in, half, pair_last,
                            alpha_residual, (void*)ds, hw, B);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_conv_dgrad(int cin, int cout, const float* w, const float* g, float* dx, const float* res, int M, int H, int W, float* wt,
                   void* wtp, const float* zero_bias, hipStream_t s, int dt) {
    int rc;
    if ((rc = hrn_launch_dgrad_weights(w, wt, cin, cout, s))) return rc;
    if ((rc = hrn_launch_conv_pack(dt, cout, cin, wt, wtp, s))) return rc;               // a cout -> cin convolution
    ConvParams p = ConvParams();
    p.M = M; p.H = H; p.W = W;
    p.in = g; p.out = dx; p.wpk = wtp; p.bias = zero_bias; p.slope = nullptr;
    if (res) { p.res = res; p.res_mode = 1; }
    if (dt == HRN_BF16X3) {               // every tensor a pair of bf16 planes, the lo plane directly behind the hi plane
        const size_t px = (size_t)M * H * W;
        p.in_lo = px * cout * 2; p.out_lo = px * cin * 2; p.res_lo = px * cin * 2;
    }
    return hrn_launch_conv3x3(dt, cout, cin, p, s);
}
