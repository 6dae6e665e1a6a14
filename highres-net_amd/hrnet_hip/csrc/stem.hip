// Reference frame (per-pixel lower median) and the 2->64 stem convolution.
//
//   median : /root/reference/src/DeepNetworks/HRNet.py:200   torch.median(lrs[:, :9], 1)  (lower middle, pads included)
//   stem   : HRNet.py:201-204 (repeat / cat / view are never materialised) + :51-53 conv(2->64)+PReLU
//            ShiftNet.py:58 (per-plane mean subtraction) + :16 conv(2->64)   (BatchNorm/ReLU applied by bn_act kernel)
//
// Both are HBM-bound (stem: 8 B read, 128/256 B written per pixel); the conv runs on the fp32 VALU so the raw
// uint16-range image values never pass through bf16.
#include "kernels.h"

namespace {

__device__ __forceinline__ void cswap(float& a, float& b) {
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo; b = hi;
}

// lrs [B][V][HW] f32 -> ref [B][HW] f32
__global__ void median_kernel(const float* __restrict__ lrs, float* __restrict__ ref, int V, size_t hw, size_t total) {
    const int n = V < 9 ? V : 9;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t b = idx / hw, pix = idx - b * hw;
        const float* src = lrs + b * V * hw + pix;
        float v[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) v[i] = i < n ? src[(size_t)i * hw] : __builtin_inff();
        // optimal 25-exchange / 7-layer sorting network for 9 keys (verified exhaustively by the 0-1 principle in
        // tests/test_host_logic.py); +inf padding sinks to the end so index (n-1)/2 is the lower median of n keys
        cswap(v[0], v[3]); cswap(v[1], v[7]); cswap(v[2], v[5]); cswap(v[4], v[8]);
        cswap(v[0], v[7]); cswap(v[2], v[4]); cswap(v[3], v[8]); cswap(v[5], v[6]);
        cswap(v[0], v[2]); cswap(v[1], v[3]); cswap(v[4], v[5]); cswap(v[7], v[8]);
        cswap(v[1], v[4]); cswap(v[3], v[6]); cswap(v[5], v[7]);
        cswap(v[0], v[1]); cswap(v[2], v[4]); cswap(v[3], v[5]); cswap(v[6], v[8]);
        cswap(v[2], v[3]); cswap(v[4], v[5]); cswap(v[6], v[7]);
        cswap(v[1], v[2]); cswap(v[3], v[4]); cswap(v[5], v[6]);
        const int k = (n - 1) >> 1;     // lower median
        float out = v[0];
#pragma unroll
        for (int i = 1; i < 9; ++i) out = (i == k) ? v[i] : out;
        ref[idx] = out;
    }
}

// One thread = 4 pixels in a COLUMN (rows y0..y0+3, one x) x 8 output channels; lane & 7 = channel group, so 8
// consecutive lanes write one pixel's 64 channels (128 B in bf16) and a wave instruction writes 8 consecutive pixels:
// every store is a run of full 128-byte lines (the first version gave each lane 16 channels of 4 adjacent pixels and
// wrote 16 partial lines per instruction: 1.56 ms at c3, 24 % of the HBM rate).  Block = 256 threads = 32 x-positions x
// 8 channel groups = a 4 x 32 pixel patch; the 3 x 6 x 2 input window of a thread is shared along x through L1.
// weights w [64][2][3][3] f32 (OIHW) staged transposed in LDS as wl[ci*9+tap][64].
template <int DT>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                   size_t img_stride0, int rep1, size_t img_stride1,
                                                   const float* __restrict__ sub,   // [M][2] per-plane means or null
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   const float* __restrict__ slope, void* __restrict__ out,
                                                   int M, int H, int W, const float* __restrict__ only_if_nonpos, size_t out_lo) {
    if (only_if_nonpos && only_if_nonpos[0] > 0.f) return;      // (ConvParams::only_if_nonpos)
    __shared__ __attribute__((aligned(16))) float wl[18 * 64];
    __shared__ __attribute__((aligned(16))) float bl[64];
    for (int i = threadIdx.x; i < 18 * 64; i += 256) {
        const int co = i & 63, k = i >> 6;           // k = ci*9 + tap
        wl[i] = w[co * 18 + k];
    }
    if (threadIdx.x < 64) bl[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const float a = slope ? slope[0] : 1.f;
    const int cg = threadIdx.x & 7;                  // channel group: couts cg*8 .. +7
    const int xl = threadIdx.x >> 3;                 // 0..31
    const int px_tiles = (W + 31) >> 5, py_tiles = (H + 3) >> 2;
    const size_t patches = (size_t)M * py_tiles * px_tiles;
    for (size_t pi = blockIdx.x; pi < patches; pi += gridDim.x) {
        const int tx = (int)(pi % px_tiles);
        const int ty = (int)((pi / px_tiles) % py_tiles);
        const int m = (int)(pi / ((size_t)px_tiles * py_tiles));
        const int x = tx * 32 + xl, y0 = ty * 4;
        const float* p0 = in0 + (size_t)m * img_stride0;
        const float* p1 = in1 + (size_t)(m / rep1) * img_stride1;
        const float s0 = sub ? sub[2 * m] : 0.f, s1 = sub ? sub[2 * m + 1] : 0.f;
        float win[2][6][3];
#pragma unroll
        for (int dy = 0; dy < 6; ++dy) {
            const int gy = y0 + dy - 1;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int gx = x + dx - 1;
                const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                win[0][dy][dx] = ok ? p0[(size_t)gy * W + gx] - s0 : 0.f;
                win[1][dy][dx] = ok ? p1[(size_t)gy * W + gx] - s1 : 0.f;
            }
        }
        float acc[4][8];
#pragma unroll
        for (int py = 0; py < 4; ++py)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[py][c] = bl[cg * 8 + c];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* wr = wl + (ci * 9 + dy * 3 + dx) * 64 + cg * 8;
                    const f32x4 w0 = *(const f32x4*)(wr), w1 = *(const f32x4*)(wr + 4);
#pragma unroll
                    for (int py = 0; py < 4; ++py) {
                        const float xin = win[ci][py + dy][dx];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[py][j] = fmaf(xin, w0[j], acc[py][j]);
                            acc[py][4 + j] = fmaf(xin, w1[j], acc[py][4 + j]);
                        }
                    }
                }
        if (x < W) {
#pragma unroll
            for (int py = 0; py < 4; ++py) {
                if (y0 + py >= H) continue;
                const size_t o = (((size_t)m * H + y0 + py) * W + x) * 64 + cg * 8;
                f32x4 v0, v1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t0 = acc[py][j], t1 = acc[py][4 + j];
                    v0[j] = t0 >= 0.f ? t0 : a * t0;
                    v1[j] = t1 >= 0.f ? t1 : a * t1;
                }
                if constexpr (DT == HRN_BF16X3) {         // the fp32 result as (hi, lo) bf16 planes, lo plane out_lo bytes further on
                    unsigned h[4], l[4];
                    split2_bf16(v0[0], v0[1], h[0], l[0]); split2_bf16(v0[2], v0[3], h[1], l[1]);
                    split2_bf16(v1[0], v1[1], h[2], l[2]); split2_bf16(v1[2], v1[3], h[3], l[3]);
                    const u32x4 uh = {h[0], h[1], h[2], h[3]}, ul = {l[0], l[1], l[2], l[3]};
                    *(u32x4*)((unsigned short*)out + o) = uh;
                    *(u32x4*)((unsigned char*)out + out_lo + o * 2) = ul;
                } else if constexpr (DT == HRN_BF16) {
                    u32x4 u;
                    u[0] = pack2_bf16(v0[0], v0[1]); u[1] = pack2_bf16(v0[2], v0[3]);
                    u[2] = pack2_bf16(v1[0], v1[1]); u[3] = pack2_bf16(v1[2], v1[3]);
                    *(u32x4*)((unsigned short*)out + o) = u;
                } else {
                    *(f32x4*)((float*)out + o) = v0;
                    *(f32x4*)((float*)out + o + 4) = v1;
                }
            }
        }
    }
}

// bf16 storage path: the same 2->64 conv on the matrix cores.  K = 18 is tiny, but the VALU version above issues ~1000
// instructions per 4 pixels x 8 channels and ran at 1.65 ms (24 % of the HBM rate) at c3.  Here each input value is
// split into two bf16 (hi = bf16(v), lo = bf16(v - hi): ~16 significant bits, so the uint16-range image is not rounded
// to 8) and the K axis becomes [18 hi | 18 lo | 12 zero] = 48 = three 32x32x16 k-steps with the (bf16) weights repeated
// for the lo half: D[co][px] = W hi + W lo.  One wave = one 32-pixel row segment x 64 channels = 6 MFMAs; fragments are
// built in registers straight from the global loads (no LDS for operands); the result is transposed through a
// wave-private LDS tile so that the NHWC stores are 16 B per lane, 8 lanes per 128-byte pixel line.
// X3 (the bf16x3 mode): the WEIGHTS are split as well and the third term W lo x X hi joins the K axis - [18 hi | 18 hi | 18 lo | 10 zero]
// against [18 hi | 18 lo | 18 hi | 10 zero] = 64 = four k-steps: fp32-grade products (~2^-16) -, and the fp32 result leaves as a pair of
// bf16 planes (hi, lo), the lo plane out_lo bytes behind the hi plane.
template <bool X3>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                        size_t img_stride0, int rep1, size_t img_stride1,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ slope, unsigned short* __restrict__ out,
                                                        int M, int H, int W, size_t out_lo) {
    constexpr int SROW = 144;                                        // staged pixel row: 64 bf16 + 16 B pad
    constexpr int NK = X3 ? 4 : 3;                                   // k-steps of 16
    __shared__ __attribute__((aligned(16))) unsigned char stg_all[(X3 ? 2 : 1) * 4 * 32 * SROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (uniform: the segment arithmetic below stays on the scalar unit)
    const int r = lane & 31, hh = lane >> 5;
    const float a = slope ? slope[0] : 1.f;
    unsigned char* stg = stg_all + wave * 32 * SROW;

    // K layout (round 3): the two 32-lane halves of the wave take one INPUT CHANNEL each (hh = 0: the view, hh = 1: the reference frame),
    // so a lane loads and splits only its own 9 taps, and nothing is selected by hh afterwards.  Slot n = 8 s + j of lane (r, hh):
    //   B (pixel r):  n 0..8 hi[n] | 9..17 lo[n - 9] | 18 the constant 1 | X3: 19..27 hi[n - 19] | else 0
    //   A (cout r):   n 0..8 Wh[hh][n] | 9..17 Wh[hh][n - 9] | 18 bias: its bf16 hi part in the hh = 0 half, its lo part in the other |
    //                 X3: 19..27 Wl[hh][n - 19] | else 0          (Wh = bf16(w), Wl = bf16(w - Wh))
    // i.e. W hi x X (hi + lo) (+ W lo x X hi) + bias in one accumulation: fp32-grade products, no bias add in the epilogue.
    // (Until round 3 every lane converted all 18 taps and picked 12 of the 24 by hh: ~330 VALU per 32 pixels, and the kernel sat at
    // 3.7 TB/s of writes where a fill reaches 6.7.)
    bf16x8 wa[2][NK];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int s = 0; s < NK; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = 8 * s + j, co = cb * 32 + r;
                float v = 0.f;
                bool want_lo = false;
                if (n < 18) v = w[co * 18 + hh * 9 + (n % 9)];
                else if (n == 18) { v = bias[co]; want_lo = hh != 0; }
                else if (X3 && n < 28) { v = w[co * 18 + hh * 9 + (n - 19)]; want_lo = true; }
                const __bf16 h = (__bf16)v;
                wa[cb][s][j] = want_lo ? (__bf16)(v - (float)h) : h;
            }
    const float act_pick = a <= 1.f ? __builtin_inff() : -__builtin_inff();      // PReLU(t) = median(t, a t, +-inf)

    // segments (image m, row y, 32-pixel run sx), walked with a stride of gridDim.x * 4: the position is advanced by the stride's own
    // (dm, dy, dsx) with carries - no division per segment (the launcher keeps M * H * segs_x below 2^31)
    const int segs_x = (W + 31) >> 5;
    const unsigned nseg = (unsigned)M * (unsigned)H * (unsigned)segs_x;
    const unsigned seg_step = gridDim.x * 4u;
    const int d_sx = (int)(seg_step % (unsigned)segs_x), d_y = (int)((seg_step / (unsigned)segs_x) % (unsigned)H);
    const int d_m = (int)(seg_step / ((unsigned)segs_x * (unsigned)H));
    auto advance = [&](int& m, int& y, int& sx) __attribute__((always_inline)) {
        sx += d_sx; y += d_y; m += d_m;
        if (sx >= segs_x) { sx -= segs_x; ++y; }
        if (y >= H) { y -= H; ++m; }
    };
    // the 9 taps of this lane's channel at a segment (zeros outside the image); loaded one segment ahead
    auto load_taps = [&](int m, int y, int sx, float (&v)[9]) __attribute__((always_inline)) {
        const int x = sx * 32 + r;
        const float* pb = hh ? in1 + (size_t)(m / rep1) * img_stride1 : in0 + (size_t)m * img_stride0;
        // (predicated loads from one row base with constant offsets; clamped addresses + a select per tap were measured slower: 0.62
        // against 0.44 ms - every tap then needs its own 64-bit address)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int gy = y + dy - 1;                          // uniform
            const bool oky = (unsigned)gy < (unsigned)H;
            const float* rp = pb + (size_t)(oky ? gy : 0) * W + (x - 1);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = (oky && (unsigned)(x + dx - 1) < (unsigned)W) ? rp[dx] : 0.f;
        }
    };
    unsigned si = blockIdx.x * 4u + (unsigned)wave;
    int sx = (int)(si % (unsigned)segs_x), y = (int)((si / (unsigned)segs_x) % (unsigned)H), m = (int)(si / ((unsigned)segs_x * (unsigned)H));
    int nsx = sx, ny = y, nm = m;
    float vn[9];
    if (si < nseg) load_taps(m, y, sx, vn);
    for (; si < nseg; si += seg_step, sx = nsx, y = ny, m = nm) {
        __bf16 hi[9], lo[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const __bf16 h = (__bf16)vn[t];
            hi[t] = h;
            lo[t] = (__bf16)(vn[t] - (float)h);
        }
        advance(nm, ny, nsx);
        if (si + seg_step < nseg) load_taps(nm, ny, nsx, vn);
        f32x16 acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            bf16x8 bq;                                               // B[k = 16 s + 8 hh + j][col = this pixel]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = 8 * s + j;
                bq[j] = n < 9 ? hi[n] : (n < 18 ? lo[n - 9] : (n == 18 ? (__bf16)1.f : ((X3 && n < 28) ? hi[n - 19] : (__bf16)0.f)));
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][s], bq, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][s], bq, acc[1], 0, 0, 0);
        }
        // bias + PReLU + bf16 pack into the staging tile [pixel r][channel]
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = cb * 32 + 8 * g + 4 * hh;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = acc[cb][4 * g + j];              // (the bias came in through the K axis)
                    v[j] = __builtin_amdgcn_fmed3f(t, a * t, act_pick);
                }
                if constexpr (X3) {
                    uint2 uh, ul;
                    split2_bf16(v[0], v[1], uh.x, ul.x);
                    split2_bf16(v[2], v[3], uh.y, ul.y);
                    *(uint2*)(stg + r * SROW + co * 2) = uh;
                    *(uint2*)(stg + 4 * 32 * SROW + r * SROW + co * 2) = ul;
                } else {
                    uint2 u;
                    u.x = pack2_bf16(v[0], v[1]);
                    u.y = pack2_bf16(v[2], v[3]);
                    *(uint2*)(stg + r * SROW + co * 2) = u;
                }
            }
        // same wave reads back row-major: lane -> (pixel 8 i + lane/8, 16-byte part lane%8): 8 full lines per instruction
        unsigned short* orow = out + (((size_t)m * H + y) * W + sx * 32) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = 8 * i + (lane >> 3), part = lane & 7;
            const u32x4 v = *(const u32x4*)(stg + px * SROW + part * 16);
            if (sx * 32 + px < W) __builtin_nontemporal_store(v, (u32x4*)(orow + (size_t)px * 64 + part * 8));      // whole lines, read next from HBM
            if constexpr (X3) {
                const u32x4 vl = *(const u32x4*)(stg + 4 * 32 * SROW + px * SROW + part * 16);
                if (sx * 32 + px < W) __builtin_nontemporal_store(vl, (u32x4*)((unsigned char*)(orow + (size_t)px * 64 + part * 8) + out_lo));
            }
        }
    }
}

// per-plane mean: x [planes][hw] -> mean[planes]   (ShiftNet.py:58)
__global__ __launch_bounds__(256) void plane_mean_kernel(const float* __restrict__ x, float* __restrict__ mean, size_t hw) {
    const float* p = x + (size_t)blockIdx.x * hw;
    double s = 0.0;
    for (size_t i = threadIdx.x; i < hw; i += 256) s += (double)p[i];
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[blockIdx.x] = (float)(red[0] / (double)hw);
}

// 8 elements per thread: out = float(hi) + float(lo)
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const u32x4* __restrict__ hi, const u32x4* __restrict__ lo, float* __restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const u32x4 h = hi[i], l = lo[i];
        f32x4 a, b;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            a[2 * k] = __uint_as_float(h[k] << 16) + __uint_as_float(l[k] << 16);
            a[2 * k + 1] = __uint_as_float(h[k] & 0xffff0000u) + __uint_as_float(l[k] & 0xffff0000u);
            b[2 * k] = __uint_as_float(h[2 + k] << 16) + __uint_as_float(l[2 + k] << 16);
            b[2 * k + 1] = __uint_as_float(h[2 + k] & 0xffff0000u) + __uint_as_float(l[2 + k] & 0xffff0000u);
        }
        *(f32x4*)(out + i * 8) = a;
        *(f32x4*)(out + i * 8 + 4) = b;
    }
}

__global__ __launch_bounds__(256) void f32_to_planes_kernel(const float* __restrict__ in, u32x4* __restrict__ hi, u32x4* __restrict__ lo, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const f32x4 a = *(const f32x4*)(in + i * 8), b = *(const f32x4*)(in + i * 8 + 4);
        unsigned h[4], l[4];
        split2_bf16(a[0], a[1], h[0], l[0]); split2_bf16(a[2], a[3], h[1], l[1]);
        split2_bf16(b[0], b[1], h[2], l[2]); split2_bf16(b[2], b[3], h[3], l[3]);
        hi[i] = u32x4{h[0], h[1], h[2], h[3]};
        lo[i] = u32x4{l[0], l[1], l[2], l[3]};
    }
}

}  // namespace

int hrn_launch_median(const float* lrs, float* ref, int B, int V, int H, int W, hipStream_t stream) {
    const size_t hw = (size_t)H * W, total = (size_t)B * hw;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    HrnProfScope prof("median9", 0.0, (double)total * 4 * ((V < 9 ? V : 9) + 1), stream);
    hipLaunchKernelGGL(median_kernel, dim3(blocks), dim3(256), 0, stream, lrs, ref, V, hw, total);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_stem(int dt, const float* in0, size_t img_stride0, const float* in1, int rep1, size_t img_stride1,
                    const float* sub, const float* w, const float* bias, const float* slope, void* out,
                    int M, int H, int W, hipStream_t stream, size_t out_lo) {
    const size_t patches = (size_t)M * ((H + 3) / 4) * ((W + 31) / 32);
    const int blocks = (int)(patches < 16384 ? patches : 16384);
    const double px = (double)M * H * W;
    HrnProfScope prof(dt == HRN_BF16 ? "stem2x64_bf16" : dt == HRN_BF16X3 ? "stem2x64_bf16x3" : "stem2x64_f32", 2.0 * 18 * 64 * px, px * (4 + (double)M / rep1 / M * 4 + 64.0 * hrn_esize(dt)), stream);
    if ((dt == HRN_BF16 || dt == HRN_BF16X3) && sub == nullptr)
        HRN_CHECK((size_t)M * H * ((W + 31) / 32) < ((size_t)1 << 31), -2, "stem: %d images of %d x %d exceed the 32-bit segment count", M, H, W);
    if (dt == HRN_BF16 && sub == nullptr) {
        const size_t nseg = (size_t)M * H * ((W + 31) / 32);
        const int mblocks = (int)((nseg + 3) / 4 < 8192 ? (nseg + 3) / 4 : 8192);
        hipLaunchKernelGGL(stem_mfma_kernel<false>, dim3(mblocks), dim3(256), 0, stream, in0, in1, img_stride0, rep1, img_stride1, w, bias, slope,
                           (unsigned short*)out, M, H, W, (size_t)0);
    } else if (dt == HRN_BF16X3 && sub == nullptr) {
        HRN_CHECK(out_lo != 0, -2, "stem bf16x3: lo-plane offset missing");
        const size_t nseg = (size_t)M * H * ((W + 31) / 32);
        const int mblocks = (int)((nseg + 3) / 4 < 8192 ? (nseg + 3) / 4 : 8192);
        hipLaunchKernelGGL(stem_mfma_kernel<true>, dim3(mblocks), dim3(256), 0, stream, in0, in1, img_stride0, rep1, img_stride1, w, bias, slope,
                           (unsigned short*)out, M, H, W, out_lo);
    } else if (dt == HRN_BF16X3) {
        HRN_CHECK(out_lo != 0, -2, "stem bf16x3: lo-plane offset missing");
        hipLaunchKernelGGL(stem_kernel<HRN_BF16X3>, dim3(blocks), dim3(256), 0, stream, in0, in1, img_stride0, rep1, img_stride1, sub, w, bias, slope, out, M, H, W,
                           (const float*)nullptr, out_lo);
    } else if (dt == HRN_BF16)
        hipLaunchKernelGGL(stem_kernel<HRN_BF16>, dim3(blocks), dim3(256), 0, stream, in0, in1, img_stride0, rep1, img_stride1, sub, w, bias, slope, out, M, H, W,
                           (const float*)nullptr, (size_t)0);
    else
        hipLaunchKernelGGL(stem_kernel<HRN_F32>, dim3(blocks), dim3(256), 0, stream, in0, in1, img_stride0, rep1, img_stride1, sub, w, bias, slope, out, M, H, W,
                           (const float*)nullptr, (size_t)0);
    HRN_LAUNCH_CHECK();
    return 0;
}

// f32, no activation, and only if only_if_nonpos[0] <= 0: the stem's pre-activation for the backward of a PReLU whose slope is not positive
int hrn_launch_stem_pre(const float* in0, size_t img_stride0, const float* in1, int rep1, size_t img_stride1, const float* w,
                        const float* bias, float* out, int M, int H, int W, const float* only_if_nonpos, hipStream_t stream, int dt) {
    const size_t patches = (size_t)M * ((H + 3) / 4) * ((W + 31) / 32);
    const int blocks = (int)(patches < 16384 ? patches : 16384);
    if (dt == HRN_BF16X3) {
        hipLaunchKernelGGL(stem_kernel<HRN_BF16X3>, dim3(blocks), dim3(256), 0, stream, in0, in1, img_stride0, rep1, img_stride1, (const float*)nullptr, w, bias,
                           (const float*)nullptr, (void*)out, M, H, W, only_if_nonpos, (size_t)M * H * W * 64 * 2);
        HRN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(stem_kernel<HRN_F32>, dim3(blocks), dim3(256), 0, stream, in0, in1, img_stride0, rep1, img_stride1, (const float*)nullptr, w, bias,
                       (const float*)nullptr, (void*)out, M, H, W, only_if_nonpos, (size_t)0);
    HRN_LAUNCH_CHECK();
    return 0;
}

// bf16x3: (hi, lo) bf16 planes -> f32 (the fused state before the fp32 decoder; staged tensors for the tests)
int hrn_launch_planes_to_f32(const void* hi, size_t lo_off, float* out, size_t n, hipStream_t stream) {
    HRN_CHECK(n % 8 == 0 && lo_off % 16 == 0, -2, "planes_to_f32: %zu elements / lo offset %zu not aligned", n, lo_off);
    const size_t n8 = n / 8;
    const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    HrnProfScope prof("planes_to_f32", 0.0, (double)n * 8, stream);
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3(blocks), dim3(256), 0, stream, (const u32x4*)hi, (const u32x4*)((const unsigned char*)hi + lo_off), out, n8);
    HRN_LAUNCH_CHECK();
    return 0;
}

// f32 -> (hi, lo) bf16 planes
int hrn_launch_f32_to_planes(const float* in, void* hi, size_t lo_off, size_t n, hipStream_t stream) {
    HRN_CHECK(n % 8 == 0 && lo_off % 16 == 0, -2, "f32_to_planes: %zu elements / lo offset %zu not aligned", n, lo_off);
    const size_t n8 = n / 8;
    const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    hipLaunchKernelGGL(f32_to_planes_kernel, dim3(blocks), dim3(256), 0, stream, in, (u32x4*)hi, (u32x4*)((unsigned char*)hi + lo_off), n8);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_plane_mean(const float* x, float* mean, int planes, size_t hw, hipStream_t stream) {
    HrnProfScope prof("plane_mean", 0.0, (double)planes * hw * 4, stream);
    hipLaunchKernelGGL(plane_mean_kernel, dim3(planes), dim3(256), 0, stream, x, mean, hw);
    HRN_LAUNCH_CHECK();
    return 0;
}
