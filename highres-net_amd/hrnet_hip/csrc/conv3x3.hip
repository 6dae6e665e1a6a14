// conv3x3 (pad 1, stride 1) + bias + PReLU + residual as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces every nn.Conv2d(k=3) of the reference hot path:
//   /root/reference/src/DeepNetworks/HRNet.py:17-22 (ResidualBlock), :51-60 (Encoder), :93-97 (fuse)
//   /root/reference/src/DeepNetworks/ShiftNet.py:19-42 (layers 2..8)
// including the data movement the reference materialises around them (slice / flip / cat of the view
// stack, HRNet.py:114-119; residual add :33; alpha residual :123-128).
//
// GEMM view (per image):  D[cout][pixel] = sum_k  Wt[cout][k] * X[k][pixel],  k = (tap, cin)
//   A operand = weights  (rows = output channels),  B operand = activations (cols = pixels)
//   -> each lane ends up with 4 consecutive output channels of ONE pixel per accumulator quad, which is the
//      NHWC store order (8 B for bf16, 16 B for f32 per store).
//
// Workgroup = 256 threads = 4 waves, output tile 8 rows x 32 cols x COUT; wave w owns tile rows 2w, 2w+1
// (two 32-pixel column blocks) and all output channels, 64 at a time ("half").
// LDS: input halo tile [10][34] pixels x 128 B of channels (+16 B pad per pixel: conflict-free ds_read_b128),
//      two weight slices [64 cout][128 B of K] (+16 B pad per row), double buffered.  67,392 B -> 2 workgroups/CU.
// K is walked in "steps": (channel chunk of 128 B) x (tap) x (64-cout half); per step a wave issues
//   bf16: 4 k-steps  x 4 MFMA 32x32x16   |  f32: 4 groups x 4 x 4 MFMA 32x32x2 (exact fp32, k-order permuted
//   identically for A and B: lane half hh supplies channel 8q+4hh+j at MFMA j of group q).
#include "conv3x3.h"

namespace {

constexpr int HALO_H = CONV_TILE_H + 2;
constexpr int HALO_W = CONV_TILE_W + 2;
constexpr int PIX_PITCH = 144;
constexpr int IN_LDS_BYTES = HALO_H * HALO_W * PIX_PITCH;   // 48,960
constexpr int W_ROW_PITCH = 144;
constexpr int W_BUF_BYTES = 64 * W_ROW_PITCH;               // 9,216
constexpr int LDS_BYTES = IN_LDS_BYTES + 2 * W_BUF_BYTES;   // 67,392
constexpr int N_IN_CHUNKS16 = HALO_H * HALO_W * 8;          // 2,720 16-byte pieces

template <int DT, int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* in_lds = smem;
    unsigned char* w_lds = smem + IN_LDS_BYTES;

    constexpr int ES = ElemOf<DT>::size;
    constexpr int KB = 128 / ES;            // channels per 128-byte chunk
    constexpr int NCHUNK = CIN / KB;
    constexpr int NHALF = COUT / 64;
    constexpr int NSTEP = NCHUNK * 9 * NHALF;

    // ---- XCD-aware tile assignment: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
    //      range of tiles so that halo rows of neighbouring tiles hit the same L2.  Bijective for any grid size.
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tiles_x = (p.W + CONV_TILE_W - 1) / CONV_TILE_W;
    const int tiles_y = (p.H + CONV_TILE_H - 1) / CONV_TILE_H;
    const int tiles = tiles_x * tiles_y;
    const int m = logical / tiles;
    const int t = logical - m * tiles;
    const int ty = t / tiles_x;
    const int y0 = ty * CONV_TILE_H, x0 = (t - ty * tiles_x) * CONV_TILE_W;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int H = p.H, W = p.W;
    const size_t hw = (size_t)H * W;

    // ---- input sources
    const unsigned char* src0 = nullptr;    // pair descriptor: view i and its partner in the stack
    const unsigned char* src1 = nullptr;
    if (p.pair_h > 0) {
        const int b = m / p.pair_h, i = m - b * p.pair_h;
        src0 = (const unsigned char*)p.stack + ((size_t)b * p.pair_vs + i) * hw * 64 * ES;
        src1 = (const unsigned char*)p.stack + ((size_t)b * p.pair_vs + (p.pair_last - i)) * hw * 64 * ES;
    }
    const bool in_pair = p.in_pair != 0;
    const unsigned char* in_plain = (const unsigned char*)p.in + (size_t)m * hw * CIN * ES;
    const int in_pix_bytes = in_pair ? 64 * ES : CIN * ES;

    auto stage_input = [&](int chunk) {
        const unsigned char* base;
        int choff;
        if (in_pair) {
            const int ch0 = chunk * KB;                 // first channel of this chunk in the virtual 128-ch input
            base = ch0 < 64 ? src0 : src1;
            choff = (ch0 & 63) * ES;
        } else {
            base = in_plain;
            choff = chunk * 128;
        }
#pragma unroll
        for (int it = 0; it < (N_IN_CHUNKS16 + 255) / 256; ++it) {
            const int c = tid + it * 256;
            if (c < N_IN_CHUNKS16) {
                const int pix = c >> 3, part = c & 7;
                const int py = pix / HALO_W, px = pix - py * HALO_W;
                const int gy = y0 + py - 1, gx = x0 + px - 1;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                    v = *(const uint4*)(base + ((size_t)gy * W + gx) * in_pix_bytes + choff + part * 16);
                *(uint4*)(in_lds + pix * PIX_PITCH + part * 16) = v;
            }
        }
    };
    const uint4* wg = (const uint4*)p.wpk;
    const int w_dst0 = (tid >> 3) * W_ROW_PITCH + (tid & 7) * 16;
    const int w_dst1 = w_dst0 + 32 * W_ROW_PITCH;

    f32x16 acc[NHALF][2][2];
#pragma unroll
    for (int h = 0; h < NHALF; ++h)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[h][a][b][e] = 0.f;

    // ---- prologue: input chunk 0 + weight slice 0
    stage_input(0);
    {
        const uint4 w0 = wg[tid], w1 = wg[tid + 256];
        *(uint4*)(w_lds + w_dst0) = w0;
        *(uint4*)(w_lds + w_dst1) = w1;
    }
    __syncthreads();

    const unsigned char* a_base = w_lds + r * W_ROW_PITCH + hh * 16;                               // + buf, + cb*32 rows
    const unsigned char* b_base = in_lds + ((2 * wave) * HALO_W + r) * PIX_PITCH + hh * 16;        // + tap, + pb row

    int step = 0;
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
        if (chunk > 0) {            // previous chunk's last step ended with a barrier: the tile is free
            stage_input(chunk);
            __syncthreads();
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const unsigned char* xb = b_base + (ky * HALO_W + kx) * PIX_PITCH;
#pragma unroll
            for (int half = 0; half < NHALF; ++half) {
                const bool more = step + 1 < NSTEP;
                uint4 w0, w1;
                if (more) {         // prefetch the next weight slice into registers
                    w0 = wg[(size_t)(step + 1) * 512 + tid];
                    w1 = wg[(size_t)(step + 1) * 512 + tid + 256];
                }
                const unsigned char* wb = a_base + (step & 1) * W_BUF_BYTES;
                if constexpr (DT == HRN_BF16) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const bf16x8 a0 = *(const bf16x8*)(wb + ks * 32);
                        const bf16x8 a1 = *(const bf16x8*)(wb + 32 * W_ROW_PITCH + ks * 32);
                        const bf16x8 b0 = *(const bf16x8*)(xb + ks * 32);
                        const bf16x8 b1 = *(const bf16x8*)(xb + HALO_W * PIX_PITCH + ks * 32);
                        acc[half][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[half][0][0], 0, 0, 0);
                        acc[half][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[half][0][1], 0, 0, 0);
                        acc[half][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[half][1][0], 0, 0, 0);
                        acc[half][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[half][1][1], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 a0 = *(const f32x4*)(wb + q * 32);
                        const f32x4 a1 = *(const f32x4*)(wb + 32 * W_ROW_PITCH + q * 32);
                        const f32x4 b0 = *(const f32x4*)(xb + q * 32);
                        const f32x4 b1 = *(const f32x4*)(xb + HALO_W * PIX_PITCH + q * 32);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[half][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[half][0][0], 0, 0, 0);
                            acc[half][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[half][0][1], 0, 0, 0);
                            acc[half][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[half][1][0], 0, 0, 0);
                            acc[half][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[half][1][1], 0, 0, 0);
                        }
                    }
                }
                if (more) {         // the other buffer was last read in step-1, which ended with a barrier
                    unsigned char* wd = w_lds + ((step + 1) & 1) * W_BUF_BYTES;
                    *(uint4*)(wd + w_dst0) = w0;
                    *(uint4*)(wd + w_dst1) = w1;
                }
                __syncthreads();
                ++step;
            }
        }
    }

    // ---- epilogue: scale/bias, PReLU/ReLU, residual, NHWC store (4 consecutive channels per lane and quad)
    size_t out_img;
    float alpha = 1.f;
    const unsigned char* res3 = nullptr;
    if (p.out_h > 0) {
        const int b = m / p.out_h, i = m - b * p.out_h;
        out_img = (size_t)b * p.out_vs + i;
        if (p.res_mode == 3) {
            if (p.alphas) alpha = p.alphas[(size_t)b * p.alpha_vs + (p.pair_last - i)];
            res3 = (const unsigned char*)p.res + ((size_t)b * p.res_vs + i) * hw * COUT * ES;
        }
    } else {
        out_img = (size_t)m;
    }
    unsigned char* outp = (unsigned char*)p.out + out_img * hw * COUT * ES;
    const bool has_slope = p.slope != nullptr;
    const float slope = has_slope ? p.slope[0] : 0.f;
    const int gx = x0 + r;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int gy = y0 + 2 * wave + pb;
        if (gy >= H || gx >= W) continue;
        const size_t pix = (size_t)gy * W + gx;
#pragma unroll
        for (int half = 0; half < NHALF; ++half) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = half * 64 + cb * 32 + 8 * g + 4 * hh;
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[half][pb][cb][4 * g + j];
                    if (p.scale) {
                        const f32x4 sc = *(const f32x4*)(p.scale + co);
                        v *= sc;
                    }
                    v += *(const f32x4*)(p.bias + co);
                    if (has_slope) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] >= 0.f ? v[j] : slope * v[j];
                    }
                    if (p.relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    if (p.res_mode == 1) {
                        v += load4<DT>((const unsigned char*)p.res + (size_t)m * hw * COUT * ES, pix * COUT + co);
                    } else if (p.res_mode == 2) {   // pair gather: channels 0..63 = view i, 64..127 = its partner
                        v += load4<DT>(half == 0 ? src0 : src1, pix * 64 + (co & 63));
                    } else if (p.res_mode == 3) {   // x_i + alpha_partner * f   (HRNet.py:127); may be in place
                        v = load4<DT>(res3, pix * COUT + co) + alpha * v;
                    }
                    store4<DT>(outp, pix * COUT + co, v);
                }
            }
        }
    }
}

template <int DT, int CIN, int COUT>
int launch(const ConvParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        HRN_HIP(hipFuncSetAttribute((const void*)conv3x3_kernel<DT, CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    const long tiles = (long)((p.W + CONV_TILE_W - 1) / CONV_TILE_W) * ((p.H + CONV_TILE_H - 1) / CONV_TILE_H);
    const long nwg = tiles * p.M;
    HRN_CHECK(nwg > 0 && nwg < (1L << 31), -2, "conv3x3: bad grid (%ld workgroups)", nwg);
    static const char* fam_names[2][2][2] = {{{"conv3x3_f32_64x64", "conv3x3_f32_64x128"}, {"conv3x3_f32_128x64", "conv3x3_f32_128x128"}},
                                             {{"conv3x3_bf16_64x64", "conv3x3_bf16_64x128"}, {"conv3x3_bf16_128x64", "conv3x3_bf16_128x128"}}};
    const double px = (double)p.M * p.H * p.W, es = ElemOf<DT>::size;
    HrnProfScope prof(fam_names[DT][CIN / 128][COUT / 128], 2.0 * CIN * COUT * 9 * px,
                      px * es * (CIN + COUT + (p.res_mode ? COUT : 0)), stream);
    hipLaunchKernelGGL((conv3x3_kernel<DT, CIN, COUT>), dim3((unsigned)nwg), dim3(256), LDS_BYTES, stream, p);
    HRN_LAUNCH_CHECK();
    return 0;
}

// ---- weight packing: OIHW f32 -> [step][64 cout][128 B of K], step = (chunk*9 + tap)*NHALF + half
template <int DT>
__global__ void conv_pack_kernel(const float* __restrict__ w, void* __restrict__ out, int cin, int cout) {
    constexpr int ES = ElemOf<DT>::size;
    constexpr int KB = 128 / ES;
    const int nhalf = cout / 64;
    const size_t total = (size_t)cin * cout * 9;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(idx % KB);
        const int col = (int)((idx / KB) % 64);
        const int step = (int)(idx / ((size_t)KB * 64));
        const int half = step % nhalf;
        const int ct = step / nhalf;
        const int tap = ct % 9, chunk = ct / 9;
        const int co = half * 64 + col, ci = chunk * KB + kk;
        const float v = w[((size_t)co * cin + ci) * 9 + tap];
        store_elem<DT>(out, idx, v);
    }
}

}  // namespace

int hrn_launch_conv3x3(int dt, int cin, int cout, const ConvParams& p, hipStream_t stream) {
    HRN_CHECK(p.M > 0 && p.H > 0 && p.W > 0, -2, "conv3x3: empty problem M=%d H=%d W=%d", p.M, p.H, p.W);
    HRN_CHECK(!p.in_pair || (cin == 128 && p.pair_h > 0 && p.stack), -2, "conv3x3: pair input needs cin=128 and a pair descriptor");
    HRN_CHECK(p.res_mode != 2 || (p.pair_h > 0 && p.stack && cout == 128), -2, "conv3x3: res_mode 2 needs a pair descriptor and cout=128");
    HRN_CHECK(p.in_pair || p.in, -2, "conv3x3: null input");
    HRN_CHECK(p.res_mode != 3 || p.out_h > 0, -2, "conv3x3: res_mode 3 needs slot output");
#define HRN_CONV_CASE(DT_, CI_, CO_) if (dt == DT_ && cin == CI_ && cout == CO_) return launch<DT_, CI_, CO_>(p, stream);
    HRN_CONV_CASE(HRN_BF16, 64, 64)
    HRN_CONV_CASE(HRN_BF16, 64, 128)
    HRN_CONV_CASE(HRN_BF16, 128, 64)
    HRN_CONV_CASE(HRN_BF16, 128, 128)
    HRN_CONV_CASE(HRN_F32, 64, 64)
    HRN_CONV_CASE(HRN_F32, 64, 128)
    HRN_CONV_CASE(HRN_F32, 128, 64)
    HRN_CONV_CASE(HRN_F32, 128, 128)
#undef HRN_CONV_CASE
    hrn_set_error("conv3x3: unsupported dtype/channels dt=%d cin=%d cout=%d", dt, cin, cout);
    return -2;
}

int hrn_launch_conv_pack(int dt, int cin, int cout, const float* w, void* packed, hipStream_t stream) {
    HRN_CHECK((cin == 64 || cin == 128) && (cout == 64 || cout == 128), -2, "conv pack: unsupported channels %d->%d", cin, cout);
    const size_t total = (size_t)cin * cout * 9;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    if (dt == HRN_BF16) hipLaunchKernelGGL(conv_pack_kernel<HRN_BF16>, dim3(blocks), dim3(256), 0, stream, w, packed, cin, cout);
    else hipLaunchKernelGGL(conv_pack_kernel<HRN_F32>, dim3(blocks), dim3(256), 0, stream, w, packed, cin, cout);
    HRN_LAUNCH_CHECK();
    return 0;
}
