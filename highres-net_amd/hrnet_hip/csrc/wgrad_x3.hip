// conv3x3 weight gradient in split-bf16 ("bf16x3"): dW[co][ci][ky][kx] += sum_pixels g[p][co] * x[p + (ky-1, kx-1)][ci]
// (nn.Conv2d 3x3 pad 1 weight gradient; `loss.backward()` through HRNet, src/train.py:190) for the bf16x3 training mode, where g and x
// are pairs of bf16 planes (hi, lo).  A product is g_hi x_hi + g_hi x_lo + g_lo x_hi on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation - the arithmetic of conv3x3_v6x3.hip with K = PIXELS: D[co][ci] += A[co][k] B[k][ci], k = 32 consecutive pixels of one
// image row.  Both operands are channels-last in memory, i.e. k is the ROW index of their LDS images: the fragments come from
// ds_read_b64_tr_b16, the hardware transpose read of gfx950 (a 4 pixel x 16 channel block per 16 lanes, column-major into the
// registers), from images whose 16-byte chunks are XOR-swizzled by the pixel (chunk ^ (2 bit1(px) ^ 4 bit3(px)): conflict-free for
// every tap offset; found by enumeration).
//
// Decomposition: one launch per (64-cout chunk, 64-cin chunk), as the fp32 kernel (backward.hip) - same partial slabs
// [workgroup][tap][64 co][64 ci], same finish kernel.  A workgroup (4 waves, two workgroups per CU) walks DOWN 32-pixel-wide column
// strips of the images: per image row it needs one new row of x (34 pixels with the halo) and one row of g, which arrive by LDS-DMA
// (buffer_load ... lds; out-of-image pixels and rows read as zeros through the descriptor's range check) two rows ahead into small
// rings - x: 5 rows, g: 3 rows, 76 KB in all - so every x row is fetched once and used for three output rows.  Wave (cb, ib) owns the
// 32 x 32 block (cout 32 cb.., cin 32 ib..) for all nine taps: 36 accumulator tiles = 144 registers, kept across the whole launch.
// Per row and wave: 8 transposed reads for g (shared by the nine taps), 8 per tap for x, 12 MFMAs per tap.
#include "kernels.h"
#include "backward.h"

#ifndef WGX_ABL
#define WGX_ABL 0       // timing-only ablations (results WRONG): 1 no DMA | 2 no transposed reads | 4 no MFMA | 8 no row barrier
#endif

namespace {

constexpr int XPX = 40, GPX = 32;                            // pixels per LDS row image (x: 34 used)
constexpr int XPLANE = XPX * 128, GPLANE = GPX * 128;        // one plane of one row: 5,120 | 4,096 B
constexpr int XSLOT = 2 * XPLANE, GSLOT = 2 * GPLANE;        // hi + lo
constexpr int XRING = 5, GRING = 3;
constexpr int OFF_G = XRING * XSLOT;                         // 51,200
constexpr int WX_LDS = OFF_G + GRING * GSLOT;                // 75,776
constexpr unsigned OOBW = 0x80000000u;

struct WgradX3Params {
    const void* x;          // plain input [M][H][W][cin] bf16 planes (in_pair == 0)
    const void* stack;      // pair gather: views [B][pair_vs][H][W][64]
    size_t x_lo;            // byte offset of the lo plane of x / stack
    int in_pair, pair_h, pair_last, pair_vs;
    const void* g;          // [M][H][W][cout] planes
    size_t g_lo;
    float* partial;         // [gridDim.x][9][64][64]
    int M, H, W, cin, cout, co_chunk, ci_chunk;
};

typedef __attribute__((address_space(3))) void* lds_ptr_w;
__device__ __forceinline__ int swz_w(int px) { return (((px >> 1) & 1) << 1) ^ (((px >> 3) & 1) << 2); }

template <int N> __device__ __forceinline__ void wait_vm_w() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(const WgradX3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = w >> 1, ib = w & 1;
    const int H = p.H, W = p.W;
    const unsigned hw = (unsigned)(H * W);
    const int strips = (W + 31) / 32;
    const long units = (long)p.M * strips;
    const unsigned xpitch = (unsigned)((p.in_pair ? 64 : p.cin) * 2), gpitch = (unsigned)(p.cout * 2);
    const unsigned xcb = p.in_pair ? 0u : (unsigned)(p.ci_chunk * 128), gcb = (unsigned)(p.co_chunk * 128);
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) unsigned char*)smem;

    f32x4 acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read addresses: lane 4q + p of a 16-lane group gq supplies row (pixel) 8 gq + 4 i + q, columns 4p..4p+3 of the 16-channel
    // block.  g: pixel k = 8 gq + 4 i + q; x, tap column kx: pixel k + kx.  Byte = pixel*128 + ((2 blk + (p >> 1)) ^ swz(pixel))*16 + 8 (p & 1)
    const int gq = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    unsigned ga[2][2], xa[3][2][2];                          // [i][block] | [kx][i][block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int px = 8 * gq + 4 * i + q4;
            ga[i][b] = lds0 + (unsigned)(OFF_G + px * 128 + (((2 * (2 * cb + b) + (p4 >> 1)) ^ swz_w(px)) << 4) + 8 * (p4 & 1));
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int pxx = px + kx;
                xa[kx][i][b] = lds0 + (unsigned)(pxx * 128 + (((2 * (2 * ib + b) + (p4 >> 1)) ^ swz_w(pxx)) << 4) + 8 * (p4 & 1));
            }
        }

    // (per image row this wave issues 6 (wave 0: x pieces 0 and 4, g piece 0; two planes each) or 4 DMA instructions)
    bool first = true;
    for (long u = blockIdx.x; u < units; u += gridDim.x) {
        const int m = (int)(u / strips);
        const int x0 = (int)(u - (long)m * strips) * 32;
        // sources: image m of g; of x the plain tensor, or view i / its partner of the pair gather (64 channels each)
        size_t xoff;
        if (p.in_pair) {
            const int b = m / p.pair_h, i = m - b * p.pair_h;
            const int v = p.ci_chunk == 0 ? i : p.pair_last - i;
            xoff = ((size_t)b * p.pair_vs + v) * hw * 128;
        } else {
            xoff = (size_t)m * hw * xpitch;
        }
        const unsigned char* xbase = (const unsigned char*)(p.in_pair ? p.stack : p.x) + xoff;
        const unsigned char* gbase = (const unsigned char*)p.g + (size_t)m * hw * gpitch;
        const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, (int)(hw * xpitch), 0x00020000);
        const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc((void*)(xbase + p.x_lo), 0, (int)(hw * xpitch), 0x00020000);
        const __amdgpu_buffer_rsrc_t rg0 = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, (int)(hw * gpitch), 0x00020000);
        const __amdgpu_buffer_rsrc_t rg1 = __builtin_amdgcn_make_buffer_rsrc((void*)(gbase + p.g_lo), 0, (int)(hw * gpitch), 0x00020000);
        // per-lane byte offsets inside an image row of this wave's DMA pieces (8 pixels x 128 B each): lane i -> pixel 8 j + (i >> 3),
        // physical chunk i & 7 = logical chunk ^ swz(pixel); outside the image -> an offset the descriptor turns into zeros
        unsigned xv[2], gv;
        {
            int lq = lane;
            asm volatile("" : "+v"(lq));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int px = 8 * (k == 0 ? w : 4) + (lq >> 3), gx = x0 - 1 + px;
                const bool ok = px < 34 && (unsigned)gx < (unsigned)W;
                xv[k] = ok ? (unsigned)gx * xpitch + xcb + (unsigned)((((lq & 7) ^ swz_w(px))) << 4) : OOBW;
            }
            const int px = 8 * w + (lq >> 3), gx = x0 + px;
            gv = gx < W ? (unsigned)gx * gpitch + gcb + (unsigned)((((lq & 7) ^ swz_w(px))) << 4) : OOBW;
        }
        // one DMA instruction of an image row: item 0 / 1 = x piece w of plane 0 / 1, 2 / 3 = g piece w of plane 0 / 1, 4 / 5 = x piece 4 of
        // plane 0 / 1 (wave 0 only).  x row `row` -> ring slot (row + 1) % XRING, g row `row` -> slot row % GRING
        auto dma_item = [&](int it, int xrow, int grow) __attribute__((always_inline)) {
            if (WGX_ABL & 1) return;
            const int pl = it & 1;
            if (it == 2 || it == 3) {
                const bool ok = grow < H;
                const unsigned soff = ok ? (unsigned)grow * (unsigned)W * gpitch : 0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(pl ? rg1 : rg0, (lds_ptr_w)(smem + OFF_G + (grow % GRING) * GSLOT + pl * GPLANE + w * 1024), 16,
                                                         ok ? gv : OOBW, soff, 0, 0);
            } else {
                if (it >= 4 && w != 0) return;
                const bool ok = (unsigned)xrow < (unsigned)H;
                const unsigned soff = ok ? (unsigned)xrow * (unsigned)W * xpitch : 0u;
                const int piece = it >= 4 ? 4 : w;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(pl ? rx1 : rx0, (lds_ptr_w)(smem + ((xrow + 1) % XRING) * XSLOT + pl * XPLANE + piece * 1024), 16,
                                                         ok ? (it >= 4 ? xv[1] : xv[0]) : OOBW, soff, 0, 0);
            }
        };
        auto dma_x = [&](int row) __attribute__((always_inline)) { dma_item(0, row, 0); dma_item(1, row, 0); dma_item(4, row, 0); dma_item(5, row, 0); };
        auto dma_g = [&](int row) __attribute__((always_inline)) { dma_item(2, 0, row); dma_item(3, 0, row); };
        // ---- prologue of the strip: x rows -1 .. 2, g rows 0 .. 1 (the previous strip's last row must be done with the rings first)
        if (!first) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        first = false;
        dma_x(-1); dma_x(0); dma_x(1); dma_g(0); dma_x(2); dma_g(1);
        for (int y = 0; y < H; ++y) {
            // x rows <= y + 1 and g row y have landed once only the newest row's pieces are outstanding
            if (!(WGX_ABL & 1)) { if (w == 0) wait_vm_w<6>(); else wait_vm_w<4>(); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(WGX_ABL & 8)) __builtin_amdgcn_s_barrier();
            // (x row y + 3 goes into the slot of row y - 2 and g row y + 2 into that of row y - 1: free since the barrier above.  Their six DMA
            // instructions are issued one behind each of the first six taps' MFMAs: issued in a block in front of them they cost the wave
            // ~100 cycles each of matrix-pipe time: 17.4 against 13.1 ms of weight-gradient time per train step, tools/wgx_abl.sh)
            // ---- one k-step of 32 pixels (image row y of the strip), nine taps
            const unsigned gs_off = (unsigned)((y % GRING) * GSLOT);
            u32x2 gf[2][2][2];                               // [plane][block][i]
            auto rd = [&](u32x2& dst, unsigned addr, int imm) __attribute__((always_inline)) {
                if (WGX_ABL & 2) asm volatile("; no read" : "=v"(dst) : "v"(addr), "n"(imm));
                else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm));
            };
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (pl == 0) rd(gf[pl][b][i], ga[i][b] + gs_off, 0); else rd(gf[pl][b][i], ga[i][b] + gs_off, GPLANE);
                    }
            u32x2 xf[2][2][2][2];                            // [buffer][plane][block][i]
            auto rd_tap = [&](int buf, int t) __attribute__((always_inline)) {
                const int ky = t / 3, kx = t - 3 * ky;
                const unsigned xs_off = (unsigned)(((y + ky) % XRING) * XSLOT);          // image row y + ky - 1 lives in slot (y + ky) % XRING
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if (pl == 0) rd(xf[buf][pl][b][i], xa[kx][i][b] + xs_off, 0); else rd(xf[buf][pl][b][i], xa[kx][i][b] + xs_off, XPLANE);
                        }
            };
            rd_tap(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int buf = t & 1;
                if (t + 1 < 9) rd_tap(buf ^ 1, t + 1);
                // LDS reads return in order: with the next tap's 8 reads allowed outstanding, this tap's (and the row's g) have arrived
                if (t + 1 < 9) {
                    asm volatile("s_waitcnt lgkmcnt(8)"
                                 : "+v"(xf[buf][0][0][0]), "+v"(xf[buf][0][0][1]), "+v"(xf[buf][0][1][0]), "+v"(xf[buf][0][1][1]),
                                   "+v"(xf[buf][1][0][0]), "+v"(xf[buf][1][0][1]), "+v"(xf[buf][1][1][0]), "+v"(xf[buf][1][1][1]),
                                   "+v"(gf[0][0][0]), "+v"(gf[0][0][1]), "+v"(gf[0][1][0]), "+v"(gf[0][1][1]),
                                   "+v"(gf[1][0][0]), "+v"(gf[1][0][1]), "+v"(gf[1][1][0]), "+v"(gf[1][1][1]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(xf[buf][0][0][0]), "+v"(xf[buf][0][0][1]), "+v"(xf[buf][0][1][0]), "+v"(xf[buf][0][1][1]),
                                   "+v"(xf[buf][1][0][0]), "+v"(xf[buf][1][0][1]), "+v"(xf[buf][1][1][0]), "+v"(xf[buf][1][1][1]));
                }
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    ah[b] = __builtin_bit_cast(bf16x8, u32x4{gf[0][b][0][0], gf[0][b][0][1], gf[0][b][1][0], gf[0][b][1][1]});
                    al[b] = __builtin_bit_cast(bf16x8, u32x4{gf[1][b][0][0], gf[1][b][0][1], gf[1][b][1][0], gf[1][b][1][1]});
                    bh[b] = __builtin_bit_cast(bf16x8, u32x4{xf[buf][0][b][0][0], xf[buf][0][b][0][1], xf[buf][0][b][1][0], xf[buf][0][b][1][1]});
                    bl[b] = __builtin_bit_cast(bf16x8, u32x4{xf[buf][1][b][0][0], xf[buf][1][b][0][1], xf[buf][1][b][1][0], xf[buf][1][b][1][1]});
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        if (WGX_ABL & 4) { asm volatile("" : "+v"(acc[t][a][b]) : "v"(ah[a]), "v"(al[a]), "v"(bh[b]), "v"(bl[b])); continue; }
                        acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bh[b], acc[t][a][b], 0, 0, 0);
                        acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bl[b], acc[t][a][b], 0, 0, 0);
                        acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[a], bh[b], acc[t][a][b], 0, 0, 0);
                    }
                if (t < 6) dma_item(t, y + 3, y + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        wait_vm_w<0>();                                      // the rows fetched beyond the image (zeros) must not land in the next strip's rings
    }
    // partial[blk][tap][co 64][ci 64]: element e of lane (gq, c = lane & 15) of tile (a, b) is co = 32 cb + 16 a + 4 gq + e, ci = 32 ib + 16 b + c
    float* out = p.partial + (size_t)blockIdx.x * 9 * 4096;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    out[(size_t)t * 4096 + (32 * cb + 16 * a + 4 * gq + e) * 64 + 32 * ib + 16 * b + (lane & 15)] = acc[t][a][b][e];
}

}  // namespace

// dw[co][ci][3][3] += the weight gradient; x / g are bf16x3 plane pairs (x_lo / g_lo: byte offsets of their lo planes).  Scratch and the
// fixed-order finish are the fp32 kernel's (hrn_bwd_scratch_bytes covers 2 workgroups per CU).
int hrn_launch_conv_wgrad_x3(const void* x, const void* stack, size_t x_lo, int in_pair, int pair_h, int pair_last, int pair_vs, const void* g,
                             size_t g_lo, int M, int H, int W, int cin, int cout, float* dw, void* scratch, int num_cus, hipStream_t s) {
    HRN_CHECK((cin == 64 || cin == 128) && (cout == 64 || cout == 128), -2, "conv_wgrad_x3: unsupported %d -> %d", cin, cout);
    HRN_CHECK(!in_pair || cin == 128, -2, "conv_wgrad_x3: the pair gather has 128 input channels");
    HRN_CHECK((long)H * W * 256 < (1L << 31), -2, "conv_wgrad_x3: image too large for 32-bit in-image offsets (H=%d W=%d)", H, W);
    { const int rc_lds = hrn_allow_lds((const void*)conv_wgrad_x3_kernel, WX_LDS); if (rc_lds) return rc_lds; }
    const long units = (long)M * ((W + 31) / 32);
    int grid = 2 * num_cus;
    if (units < grid) grid = (int)units;
    WgradX3Params p;
    p.x = x; p.stack = stack; p.x_lo = x_lo; p.in_pair = in_pair; p.pair_h = pair_h; p.pair_last = pair_last; p.pair_vs = pair_vs;
    p.g = g; p.g_lo = g_lo; p.partial = (float*)scratch; p.M = M; p.H = H; p.W = W; p.cin = cin; p.cout = cout;
    const double px = (double)M * H * W;
    for (int cc = 0; cc < cout / 64; ++cc)
        for (int ic = 0; ic < cin / 64; ++ic) {
            p.co_chunk = cc; p.ci_chunk = ic;
            {
                HrnProfScope prof("conv_wgrad_bf16x3", 2.0 * 64 * 64 * 9 * px, px * 4 * 128, s);
                hipLaunchKernelGGL(conv_wgrad_x3_kernel, dim3(grid), dim3(256), WX_LDS, s, p);
            }
            if (int rc = hrn_launch_wgrad_finish((const float*)scratch, grid, dw, cin, cc, ic, s)) return rc;
        }
    HRN_LAUNCH_CHECK();
    return 0;
}
