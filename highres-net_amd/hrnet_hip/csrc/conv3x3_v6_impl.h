// Shared body of the conv3x3_v6 kernel family (included by conv3x3_v6.hip: the bf16 instantiations of a fusion level, and by
// conv3x3_v6x3.hip: the split-bf16 "bf16x3" instantiations, a separate translation unit so that neither set perturbs the other's
// register allocation).  The design notes are in conv3x3_v6.hip; what the template parameters add:
//   CIN   128 (the fusion level: 4 chunks of 32 input channels) | 64 (the encoder's layers in bf16x3: 2 chunks, pixel pitch 128 B)
//   X3    activations and weights are PAIRS of bf16 planes (hi = bf16(v), lo = bf16(v - hi): ~16 significant bits); a product is
//         hi*hi + hi*lo + lo*hi on the same MFMA with fp32 accumulation, i.e. the K loop runs three passes per chunk:
//         (x hi, w hi), (x hi, w lo) - the halo chunk stays where it is, no DMA -, (x lo, w hi).  The epilogue forms the fp32
//         result (activation, residual = hi + lo of the residual planes), splits it again and stores both planes.
//   RESM  1 = plain residual tensor [M][H][W][COUT] (encoder ResidualBlock, COUT = 64) beside 0 / 2 / 3 of conv3x3_v6.hip.
#pragma once
#include <type_traits>
#include "conv3x3.h"

// Timing-only ablations (tools/v6_abl.sh builds scratch/x/v6_<bits>/lib.so with -DV6_ABL=<bits>; results are WRONG when set):
// 1 no MFMA | 2 no epilogue | 4 no DMA | 8 no output stores | 16 no residual loads | 32 no fragment reads | 64 residual from one cache-resident 64 KB
#ifndef V6_ABL
#define V6_ABL 0
#endif
#ifndef V6_ST_AUX
#define V6_ST_AUX 2        // cache policy of the output stores: 2 = nt (the next launch reads them from HBM anyway: A/B -0.5..-1 %)
#endif
#ifndef V6_RES_NT
#define V6_RES_NT 1         // residual loads non-temporal (read once; A/B -0.6 %)
#endif
#ifndef V6_BAL
#define V6_BAL 4          // COUT = 128: eighths of a stage's steps during which waves 4-7 run at raised priority (0 = off; COUT = 64
                          // runs without: A/B on one box, 0.454 vs 0.463 ms)
#endif

#ifdef V6_STAMP      // diagnostic build only (tools/stamps/read_v6.py): s_memtime stamps of tile 1 of the largest launch
extern __device__ unsigned long long hrn_v6_stamps[256 * 8 * 24];      // defined in conv3x3_v6.hip
#define V6_ST(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (stamp_on) st[i] = t_; } while (0)
#else
#define V6_ST(i) do {} while (0)
#endif

namespace {

constexpr int T6_H = 16, T6_W = 32;
constexpr int HW6 = T6_W + 2;                              // halo width 34
constexpr int NPIX6 = (T6_H + 2) * HW6;                    // 612 halo pixels
constexpr int N_IN6 = (NPIX6 * 64 + 1023) / 1024;          // 39 DMA pieces of 1 KB per 32-channel halo chunk
constexpr int IN_BYTES6 = N_IN6 * 1024;                    // 39,936
constexpr unsigned OOB6 = 0x80000000u;                     // byte offset no descriptor of this kernel covers

template <int COUT, bool X3 = false> struct G6 {
    static constexpr int NCB = COUT / 16;                  // cout blocks of 16 per wave
    static constexpr int NQ = NCB / 2;                     // steps per tap (2 cout blocks x 4 pixel blocks = 8 MFMAs each)
    static constexpr int TAP_BYTES = COUT * 64;            // one tap x 32 cin
    static constexpr int WST = 3 * TAP_BYTES;              // one stage: 24,576 | 12,288
    static constexpr int W_PIECES = WST / 1024;            // 24 | 12
    static constexpr int OFF_IN = 2 * WST;
    static constexpr int OFF_BIAS = OFF_IN + 2 * IN_BYTES6;
    static constexpr int ROW = COUT * 2;                   // bytes per output pixel
    static constexpr int LB = NCB * 2;                     // bytes of a pixel's row one lane holds: its NCB channels NCB*c15 .. +NCB-1
    static constexpr int OFF_FIFO = OFF_BIAS + 512;        // 4 KB per wave: round 0 of the residual, fetched by LDS-DMA under the last stage
    static constexpr int FIFO_WAVE = (LB == 16 && !X3) ? 4096 : 0;
    // X3: the tile's halo offsets (5 per lane) live in LDS instead of registers, 1,280 B per wave (they are what hipcc spills first when
    // the epilogue's two residual planes push the kernel over 256 registers, and a spilled offset is reloaded in the main loop behind a
    // vmcnt(0) that drains the DMAs in flight: measured 2.62 against 1.97 ms for the layer without residual)
    static constexpr int OFF_HOFF = OFF_FIFO + 8 * FIFO_WAVE;
    static constexpr int LDS_BYTES = OFF_HOFF + (X3 ? 8 * 5 * 256 : 0);
};

typedef __attribute__((address_space(3))) void* lds_ptr6;

template <int N> __device__ __forceinline__ void wait_vm6() {
    static_assert(N >= 0 && N <= 63, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm6_rt(int n) {       // n is wave-uniform, 0..9
    switch (n) {
        case 1: wait_vm6<1>(); break;
        case 2: wait_vm6<2>(); break;
        case 3: wait_vm6<3>(); break;
        case 4: wait_vm6<4>(); break;
        case 5: wait_vm6<5>(); break;
        case 6: wait_vm6<6>(); break;
        case 7: wait_vm6<7>(); break;
        case 8: wait_vm6<8>(); break;
        case 9: wait_vm6<9>(); break;
        default: wait_vm6<0>(); break;
    }
}
__device__ __forceinline__ void barrier6() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ int swz6(int row) { return ((row >> 2) & 1) << 1; }

template <int N, int I = 0, class F> __device__ __forceinline__ void static_for6(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for6<N, I + 1>(f); }
}

template <int CIN, int COUT, int RESM, bool PAIR, bool X3>
__global__ __launch_bounds__(512, 2) void conv3x3_v6_kernel(const ConvParams p) {
    typedef G6<COUT, X3> GEO;
    static_assert(CIN == 128 || (CIN == 64 && !PAIR), "CIN");
    constexpr int NCH = CIN / 32;                            // chunks of 32 input channels
    constexpr int NCC = X3 ? 3 * NCH : NCH;                  // chunk passes of a tile (X3: hi*hi, hi*lo, lo*hi per chunk)
    constexpr int NCB = GEO::NCB, NQ = GEO::NQ, NSTEP = 3 * NQ, WST = GEO::WST, TAP_BYTES = GEO::TAP_BYTES;
    constexpr int OFF_IN = GEO::OFF_IN, ROW = GEO::ROW, LB = GEO::LB;
    typedef typename std::conditional<LB == 16, u32x4, u32x2>::type lane_row_t;       // a lane's share of one pixel's row
    constexpr bool RES = RESM != 0;
    constexpr int BAL = COUT == 128 ? V6_BAL : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bias_lds = (float*)(smem + GEO::OFF_BIAS);
    if constexpr (X3) {     // (ConvParams::only_if_nonpos: the backward's recomputation of a pre-activation, needed only behind a PReLU slope <= 0)
        if (p.only_if_nonpos && p.only_if_nonpos[0] > 0.f) return;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c15 = lane & 15, q = lane >> 4;
    const int H = p.H, W = p.W;
    const unsigned hw = (unsigned)(H * W);
    const unsigned tiles_x = (W + T6_W - 1) / T6_W;
    const unsigned tiles_y = (H + T6_H - 1) / T6_H;
    const unsigned tiles = tiles_x * tiles_y;
    const unsigned total = tiles * (unsigned)p.M;
    const unsigned G = gridDim.x;
    const unsigned bid = blockIdx.x;
    const unsigned slot0 = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;      // each XCD walks a contiguous run of tiles
    if (slot0 >= total) return;
    const int ntl = (int)((total - slot0 + G - 1) / G);
    unsigned cur_m = slot0 / tiles, cur_t = slot0 - cur_m * tiles;
    const unsigned step_m = G / tiles, step_t = G - step_m * tiles;
    constexpr bool in_pair = PAIR;                           // the conv input is the pair gather cat(view i, partner) of the stack
    constexpr unsigned in_pitch = in_pair ? 128u : (unsigned)(CIN * 2);
    const size_t src_lo = X3 ? (in_pair ? p.stack_lo : p.in_lo) : 0;     // X3: byte offset of the input's lo plane
    const unsigned char* const src0 = (const unsigned char*)(in_pair ? p.stack : p.in);
    const unsigned img_bytes = hw * in_pitch;               // < 2^31 (checked by the launcher)

    // ---- where image m of the input lives, as byte offsets from src0: (view A, view B) for the pair gather (chunks 0-1 / 2-3),
    // else one tensor image.  (Offsets, not pointers: a select between pointers in front of make_buffer_rsrc keeps hipcc from
    // promoting ANY local of this kernel to registers - ROCm 7.2.)
    auto in_bases = [&](unsigned m, size_t& a, size_t& b) __attribute__((always_inline)) {
        if (in_pair) {
            const unsigned bb = m / (unsigned)p.pair_h, i = m - bb * (unsigned)p.pair_h;
            a = ((size_t)bb * p.pair_vs + i) * hw * 128;
            b = ((size_t)bb * p.pair_vs + (p.pair_last - i)) * hw * 128;
        } else {
            a = b = (size_t)m * hw * in_pitch;
        }
    };
    // ---- per-lane byte offsets of this wave's halo pieces for tile t (pieces j = w + 8 jj < 39; lane i -> halo pixel
    // j*16 + (i >> 2), physical 16-byte chunk i & 3 = logical chunk ^ swz6(pixel)); invalid pixels -> OOB6
    unsigned hoff[5];
    auto tile_offsets = [&](unsigned t) __attribute__((always_inline)) {
        const int ty = t / tiles_x;
        const int y0 = ty * T6_H, x0 = (t - ty * tiles_x) * T6_W;
        int lq = lane;
        asm volatile("" : "+v"(lq));                        // keep the per-piece geometry out of long-lived registers
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int pix = (w + 8 * jj) * 16 + (lq >> 2);
            const int lc = (lq & 3) ^ swz6(pix);
            const int py = pix / HW6, px = pix - py * HW6;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool ok = pix < NPIX6 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const unsigned ho = ok ? (unsigned)(gy * W + gx) * in_pitch + (unsigned)(lc * 16) : OOB6;
            if constexpr (X3) *(unsigned*)(smem + GEO::OFF_HOFF + (w * 5 + jj) * 256 + lq * 4) = ho;
            else hoff[jj] = ho;
        }
    };
    // one halo piece: chunk c (32 channels = 64 bytes of a pixel) of the image behind `rs` -> input buffer `buf`
    auto dma_halo_v = [&](__amdgpu_buffer_rsrc_t rs, int c, int buf, int jj, unsigned voff) __attribute__((always_inline)) {
        const int j = w + 8 * jj;
        if (j < N_IN6) {
            const unsigned soff = in_pair ? (unsigned)((c & 1) * 64) : (unsigned)(c * 64);
            if (!(V6_ABL & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr6)(smem + OFF_IN + buf * IN_BYTES6 + j * 1024), 16, voff, soff, 0, 0);
        }
    };
    auto hoff_lds = [&](int jj) __attribute__((always_inline)) -> unsigned {     // X3: this lane's offset of piece jj of the tile
        return *(const unsigned*)(smem + GEO::OFF_HOFF + (w * 5 + jj) * 256 + lane * 4);
    };
    auto dma_halo = [&](__amdgpu_buffer_rsrc_t rs, int c, int buf, int jj) __attribute__((always_inline)) {
        if constexpr (X3) dma_halo_v(rs, c, buf, jj, hoff_lds(jj));
        else dma_halo_v(rs, c, buf, jj, hoff[jj]);
    };
    // one weight piece of stage (c, tg): piece qq = (tap kx = qq / NCB, cout block jb = qq % NCB): 16 couts x 64 bytes;
    // lane i -> row i >> 2 of the block = cout NCB * (i >> 2) + jb (the interleave that makes a lane's accumulators a contiguous
    // piece of its pixels' rows, see the epilogue), physical chunk i & 3 = logical (i & 3) ^ swz6(i >> 2)
    constexpr unsigned W_PLANE = 9u * CIN * COUT * 2u;        // one packed weight plane (X3: hi, then lo)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (int)(W_PLANE * (X3 ? 2 : 1)), 0x00020000);
    const unsigned w_lane_off = (unsigned)((lane >> 2) * (NCB * 128) + (((lane & 3) ^ swz6(lane >> 2)) << 4));
    auto dma_w = [&](int c, int tg, int slot_, int t3, unsigned plane_off) __attribute__((always_inline)) {
        const int qq = w + 8 * t3;
        if (qq < GEO::W_PIECES) {
            const int kx = qq / NCB, jb = qq - kx * NCB;
            const unsigned soff = plane_off + (unsigned)(((c >> 1) * 9 + tg * 3 + kx) * (COUT * 128) + (c & 1) * 64 + jb * 128);
            if (!(V6_ABL & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr6)(smem + slot_ * WST + kx * TAP_BYTES + jb * 1024), 16, w_lane_off, soff, 0, 0);
        }
    };
    constexpr int NW3 = (GEO::W_PIECES + 7) / 8;            // weight pieces a wave issues per stage: up to 3 | 2
    const int n_in = w < (N_IN6 & 7) ? (N_IN6 >> 3) + 1 : (N_IN6 >> 3);      // halo pieces of this wave per chunk: 5 (wave 7: 4)

    // PReLU(x) = x >= 0 ? x : s x is max(x, s x) for s <= 1 and min(x, s x) above: the median of (x, s x, +inf | -inf), one
    // instruction for every slope and no second code path (hipcc hoists what two paths share above the branch between them - the
    // unpacked residual of every round in flight, a product per accumulator - and spills it); no activation == slope 1
    const float act_slope = p.slope ? p.slope[0] : 1.f;
    const float act_pick = act_slope <= 1.f ? __builtin_inff() : -__builtin_inff();

    // fragment addresses.  A, cout block cb: a_off + slot*WST + kx*TAP + cb*1024.  B, pixel block pxb = (row
    // pxb >> 1, column half pxb & 1): halo pixel pixb[pxb] + tg*34 + kx.
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned a_off = lds0 + (unsigned)(c15 * 64 + ((q ^ swz6(c15)) << 4));
    // B fragment addresses, one register per (halo row 2w + j, tap column kx): pixel block pxb of stage tg reads row j = (pxb >> 1)
    // + tg; its second half (pxb & 1) lies 16 pixels = 1,024 bytes further on (same swizzle: 16 is a multiple of 8).  The input
    // buffer's offset is folded in once per chunk (bsel), so a read costs no VALU at all (it was 5 per read: stamps).
    unsigned baddr[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int pix = (2 * w + j) * HW6 + c15 + kx;
            baddr[j][kx] = lds0 + (unsigned)OFF_IN + (unsigned)(pix << 6) + ((unsigned)(q << 4) ^ (unsigned)((pix & 4) << 3));
        }

    f32x4 acc[NCB][4];                                      // [cout block of 16][pixel block of 16]
#ifdef V6_STAMP
    unsigned long long st[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) st[i] = 0;
    bool stamp_on = false;
#endif

    // ---- prologue: weights of stage 0, halo chunk 0 of the first tile
    if (tid < COUT) bias_lds[tid] = p.bias[tid];
    size_t inA, inB;
    in_bases(cur_m, inA, inB);
    tile_offsets(cur_t);
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src0 + inA), 0, (int)img_bytes, 0x00020000);
#pragma unroll
        for (int t3 = 0; t3 < NW3; ++t3) dma_w(0, 0, 0, t3, 0u);
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) dma_halo(rs, 0, 0, jj);
    }
    wait_vm6<0>();
    barrier6();

    for (int tl = 0; tl < ntl; ++tl) {
        const bool more_tiles = tl + 1 < ntl;
#ifdef V6_STAMP
        stamp_on = false;
        if (RESM == 2 && !PAIR && ntl >= 32 && tl == 1) { stamp_on = true; V6_ST(20); }
        if (RESM == 2 && !PAIR && ntl >= 32 && tl == 2) { stamp_on = true; V6_ST(21); stamp_on = false; }
#endif
        unsigned nxt_t = cur_t + step_t, nxt_m = cur_m + step_m;
        if (nxt_t >= tiles) { nxt_t -= tiles; ++nxt_m; }
        size_t nxA = inA, nxB = inB;
        if (more_tiles) in_bases(nxt_m, nxA, nxB);
        // accumulators start at the bias: every element of acc[cb][.] of this lane is channel NCB * c15 + cb
#pragma unroll
        for (int cb4 = 0; cb4 < NCB; cb4 += 4) {
            const f32x4 b = *(const f32x4*)(bias_lds + NCB * c15 + cb4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int pxb = 0; pxb < 4; ++pxb) acc[cb4 + e][pxb] = f32x4{b[e], b[e], b[e], b[e]};
        }
        // geometry of this tile's outputs (used by the residual prefetch and the epilogue)
        const int ty_ = cur_t / tiles_x;
        const int y0 = ty_ * T6_H, x0 = (cur_t - ty_ * tiles_x) * T6_W;
        const unsigned char *resA = nullptr, *resB = nullptr;
        unsigned char* outp;
        float res_alpha = 1.f;
        {
            size_t oimg = cur_m;
            if (p.out_h > 0) {
                const unsigned ob = cur_m / (unsigned)p.out_h, oi = cur_m - ob * (unsigned)p.out_h;
                oimg = (size_t)ob * p.out_vs + oi;
                if (RESM == 3) {
                    resA = resB = (const unsigned char*)p.res + ((size_t)ob * p.res_vs + oi) * hw * 128;
                    if (p.alphas) res_alpha = p.alphas[(size_t)ob * p.alpha_vs + (p.pair_last - oi)];
                }
            }
            if (RESM == 2) {
                const unsigned bb = cur_m / (unsigned)p.pair_h, i = cur_m - bb * (unsigned)p.pair_h;
                resA = (const unsigned char*)p.stack + ((size_t)bb * p.pair_vs + i) * hw * 128;
                resB = (const unsigned char*)p.stack + ((size_t)bb * p.pair_vs + (p.pair_last - i)) * hw * 128;
            }
            if (RESM == 1) resA = resB = (const unsigned char*)p.res + (size_t)cur_m * hw * ROW;     // plain residual tensor [M][H][W][COUT]
            outp = (unsigned char*)p.out + oimg * hw * ROW;
        }
        // the residual of round r (= pixel block r of the wave), piece j: lane (q, c15) fetches its own share of pixel 4q + j, i.e. of
        // z = cat(view i, partner) (64 channels = 128 bytes each) the 16 bytes that hold channels 8 c15 .. 8 c15 + 7
        auto res_src = [&](int r, int j) __attribute__((always_inline)) -> const unsigned char* {
            int lq = lane;
            asm volatile("" : "+v"(lq));
            const int c15r = lq & 15, qr = lq >> 4;
            const int gy = y0 + 2 * w + (r >> 1), gyc = gy < H ? gy : H - 1;
            const int gx = x0 + 16 * (r & 1) + 4 * qr + j, gxc = gx < W ? gx : W - 1;
            const unsigned char* view = (RESM == 2 && c15r >= 8) ? resB : resA;
            constexpr int RPITCH = RESM == 1 ? ROW : 128;       // bytes per pixel of the residual: a 64-channel view of the stack, or the plain tensor
            if (V6_ABL & 64) return resA + ((unsigned)(((gyc * W + gxc) & 511) * RPITCH) + (RESM == 2 ? (unsigned)((c15r & 7) * 16) : (unsigned)(c15r * LB)));
            return view + ((unsigned)((gyc * W + gxc) * RPITCH) + (RESM == 2 ? (unsigned)((c15r & 7) * 16) : (unsigned)(c15r * LB)));
        };
        constexpr bool FIFO = RES && LB == 16 && !X3;       // (there is no 8-byte LDS-DMA: the 64-cout layer fetches round 0 like the others)
        // round 0 goes into the wave's 4 KB of LDS by DMA while the tile's last stage computes (no registers to hold it beside the
        // fragments): lane i's 16 bytes land at piece j, position i, and that is where the lane reads them back
        auto res_dma0 = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!(V6_ABL & 16))
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)res_src(0, j),
                                                     (lds_ptr6)(smem + GEO::OFF_FIFO + w * GEO::FIFO_WAVE + j * 1024), 16, 0, V6_RES_NT ? 2 : 0);
        };

        // chunk pass cc: plain: chunk c = cc, in input buffer c & 1.  X3: pass ps = cc % 3 of chunk c = cc / 3 - (x hi, w hi), (x hi, w lo),
        // (x lo, w hi) -, the hi chunk in buffer 0 (passes 0 and 1), the lo chunk in buffer 1 (pass 2)
        int c = 0, ps = 0;
        for (int cc = 0; cc < NCC; ++cc) {
            if ((cc > 0 || tl > 0) && (!X3 || ps != 1)) {      // the pass reads the other input buffer: move the B addresses over
                const unsigned d = (X3 ? ps == 2 : (c & 1) != 0) ? (unsigned)IN_BYTES6 : (unsigned)-IN_BYTES6;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) baddr[j][kx] += d;
            }
            if (cc == NCC - 1) {        // this tile's last halo chunk is on its way: from here on the DMA state describes the next tile
                inA = nxA; inB = nxB;
                if (more_tiles) tile_offsets(nxt_t);
            }
            auto stage = [&](auto tg_c) __attribute__((always_inline)) {
                constexpr int tg = decltype(tg_c)::value;
#ifdef V6_STAMP
                const bool so_ = stamp_on;
                stamp_on = so_ && (cc == 1 || (cc == NCC - 1 && tg == 2));
                const int sb_ = cc == 1 ? 4 * tg : 12;
                V6_ST(sb_ + 0);
#endif
                const int slot_r = (cc + tg) & 1;                                       // ring slot this stage reads (3 stages per pass)
                const bool have_next = cc < NCC - 1 || tg < 2 || more_tiles;            // there is a stage s+1
                // X3 pass 1 re-reads the hi chunk: no halo chunk to fetch under it
                const bool next_chunk = (!X3 || ps != 1) && (cc < NCC - 1 || more_tiles);  // a halo chunk is fetched under this pass
                // stage s+1: tap row tg2 of chunk c2 (X3: of weight plane wpl2: lo in pass 1)
                const int tg2 = (tg + 1) % 3;
                int c2 = c;
                unsigned wpl2 = 0;
                if (X3) {
                    int ps2 = ps;
                    if (tg == 2) { if (++ps2 == 3) { ps2 = 0; c2 = (c + 1) & (NCH - 1); } }
                    wpl2 = ps2 == 1 ? W_PLANE : 0u;
                } else {
                    c2 = (c + (tg + 1) / 3) & (NCH - 1);
                }
                // the next halo chunk: plain: chunk c + 1 -> buffer (c + 1) & 1.  X3: under pass 0 the lo plane of chunk c -> buffer 1,
                // under pass 2 the hi plane of chunk c + 1 -> buffer 0
                const bool nlo = X3 && ps == 0;
                const int cn = nlo ? c : (c + 1) & (NCH - 1);
                const int nbuf = X3 ? (nlo ? 1 : 0) : (cn & 1);
                size_t hb = ((in_pair && cn >= 2) ? inB : inA) + (nlo ? src_lo : (size_t)0);      // in the tile's last pass these already are the next tile's views
                if constexpr (X3) {
                    // (the image bases are wave-uniform, but where hipcc keeps them in vector registers it wraps every DMA of the stage in a
                    // waterfall loop: say so - CDNA guide T20)
                    hb = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)hb) |
                         ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(hb >> 32)) << 32);      // (the builtin returns a signed int)
                }
                const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)(src0 + hb), 0, (int)img_bytes, 0x00020000);
                // DMA item `it` of this stage, issued from the gap behind the it-th step: weights of stage s+1 first, then (tg 0: pieces
                // jj 0-2, tg 1: pieces 3-4) of the next halo chunk
                // X3: the offsets of the pieces this stage issues come from LDS, read here - nothing else of the wave is in flight on the LDS
                // queue right behind the stage barrier - and waited for at once, so that no wait lands in the fragment stream
                unsigned hst[3] = {0u, 0u, 0u};
                if constexpr (X3) {
                    if (tg < 2 && next_chunk) {
#pragma unroll
                        for (int k = 0; k < (tg == 0 ? 3 : 2); ++k) hst[k] = hoff_lds(tg == 0 ? k : 3 + k);
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hst[0]), "+v"(hst[1]), "+v"(hst[2]));
                    }
                }
                auto issue_item = [&](int it) __attribute__((always_inline)) {
                    if (it < NW3) { if (have_next) dma_w(c2, tg2, slot_r ^ 1, it, wpl2); }
                    else if (tg == 0 && it < NW3 + 3) {
                        if (next_chunk) { if constexpr (X3) dma_halo_v(rs_h, cn, nbuf, it - NW3, hst[it - NW3]); else dma_halo(rs_h, cn, nbuf, it - NW3); }
                    } else if (tg == 1 && it < NW3 + 2) {
                        if (next_chunk) { if constexpr (X3) dma_halo_v(rs_h, cn, nbuf, it - NW3 + 3, hst[it - NW3]); else dma_halo(rs_h, cn, nbuf, it - NW3 + 3); }
                    }
                };
                constexpr int N_ITEMS = NW3 + (tg == 0 ? 3 : tg == 1 ? 2 : 0);
                static_assert(N_ITEMS <= NSTEP, "one DMA item per step");
                int halo_out = 0;                                                      // halo pieces this wave leaves in flight
                if (next_chunk) halo_out = tg == 0 ? 3 : tg == 1 ? n_in - 3 : 0;
                if (FIFO && tg == 2 && cc == NCC - 1 && !(V6_ABL & 2)) res_dma0();
                // ---- NSTEP steps = 3 taps x NQ cout pairs, 8 MFMAs each ((k, pxb): cout block 2qt+k x pixel block pxb).
                // Hand-issued fragment reads with counted waits: with an LDS-DMA anywhere in a kernel hipcc stops counting LDS waits
                // and answers every fragment use with lgkmcnt(0), i.e. with the whole LDS latency.  Program order of the reads:
                // prologue B0..B3(tap 0), A0(0), A1(0); step i: A0(i+1) after MFMA 0, A1(i+1) after MFMA 1, and in the second
                // step of a tap the next tap's B0..B3 after MFMAs 2..5.  LDS reads return in order; A0(i) is younger than every
                // B of its tap, so two waits per step suffice: before MFMA 0 (A0(i)) and before MFMA 4 (A1(i)), each allowing
                // exactly the reads issued after the one it needs.
                const unsigned abase = a_off + (unsigned)(slot_r * WST);
                bf16x8 fa[2][2], fb[2][4];
                auto rd = [&](bf16x8& dst, unsigned addr, int imm) __attribute__((always_inline)) {
                    if (V6_ABL & 32) asm volatile("; no read" : "=v"(dst) : "v"(addr), "n"(imm));
                    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm));
                };
                auto load_b1 = [&](int tap, int pxb) __attribute__((always_inline)) {
                    rd(fb[tap & 1][pxb], baddr[(pxb >> 1) + tg][tap], (pxb & 1) * 1024);
                };
                auto load_a1 = [&](int i, int k) __attribute__((always_inline)) {       // step i = (tap i / NQ, cout pair i % NQ)
                    rd(fa[i & 1][k], abase, (i / NQ) * TAP_BYTES + (i % NQ) * 2048 + k * 1024);
                };
                // the SIMD's arbiter prefers the older wave (w) to its partner (w + 4) all stage long: w finishes its 96 MFMAs in ~2.7 k
                // cycles and then idles at the barrier while w + 4 runs alone at ~60 % of the pipe.  Priority for w + 4 during the first
                // steps of the stage evens them out (stamps: profiles/r02_final_v6_stamps.txt)
                if (BAL && w >= 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int pxb = 0; pxb < 4; ++pxb) load_b1(0, pxb);
                load_a1(0, 0);
                load_a1(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                static_for6<NSTEP>([&](auto i_c) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_c)::value;          // a true constant: the DMA item behind step i indexes hoff[]
                    constexpr int qt = i % NQ, tap = i / NQ, bs = tap & 1;
                    constexpr bool a_next = i + 1 < NSTEP;
                    // the step that issues the next tap's B reads: the second step of a tap (NQ >= 2), with one more tap to go
                    constexpr bool b_cur = qt == (NQ > 1 ? 1 : 0) && tap + 1 < 3;
                    constexpr bool b_prev = i >= 1 && ((i - 1) % NQ) == (NQ > 1 ? 1 : 0) && (i - 1) / NQ + 1 < 3;
                    // reads allowed to be outstanding at the two waits of a step: before MFMA 0 (needs A0(i); with NQ == 2 the B
                    // fragments issued in the previous step are needed at once: everything) and before MFMA 4 (needs A1(i))
                    constexpr int n0 = (b_prev && NQ == 2) ? 0 : 1 + (b_prev ? 4 : 0);
                    constexpr int n4 = (b_prev ? 4 : 0) + (a_next ? 2 : 0) + (b_cur ? 2 : 0);
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int k = g >> 2, pxb = g & 3;
                        if (g == 0) {
                            if (n0 == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[i & 1][0]), "+v"(fb[bs][0]), "+v"(fb[bs][1]), "+v"(fb[bs][2]), "+v"(fb[bs][3]));
                            else if (n0 == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa[i & 1][0]), "+v"(fb[bs][0]));
                            else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa[i & 1][0]), "+v"(fb[bs][0]));
                        } else if (g == 4) {
                            if (n4 == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[i & 1][1]));
                            else if (n4 == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[i & 1][1]));
                            else if (n4 == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[i & 1][1]));
                            else if (n4 == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fa[i & 1][1]));
                            else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[i & 1][1]));
                        } else if (k == 0) asm volatile("" : "+v"(fb[bs][pxb]));
                        if (V6_ABL & 1) asm volatile("" : "+v"(acc[qt * 2 + k][pxb]) : "v"(fa[i & 1][k]), "v"(fb[bs][pxb]));
                        else acc[qt * 2 + k][pxb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[bs][pxb], fa[i & 1][k], acc[qt * 2 + k][pxb], 0, 0, 0);
                        if (g < 2 && a_next) load_a1(i + 1, g);
                        if (g >= 2 && g < 6 && b_cur) load_b1(tap + 1, g - 2);
                        if (g == 7 && i < N_ITEMS) issue_item(i);
                        if (BAL && g == 7 && i == (NSTEP * BAL) / 8 - 1 && w >= 4) __builtin_amdgcn_s_setprio(0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                // stage s+1's weights (and every older DMA) have landed once only this stage's halo pieces are outstanding
                V6_ST(sb_ + 1);
                wait_vm6_rt(halo_out);
                V6_ST(sb_ + 2);                                   // tg 2: 0 (the residual prefetch is older than the weights)
                if (tg == 2 && cc == NCC - 1 && (V6_ABL & 2)) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int pxb = 0; pxb < 4; ++pxb) asm volatile("" :: "v"(acc[cb][pxb]));
                }
                if constexpr (X3) {
                if (tg == 2 && cc == NCC - 1 && !(V6_ABL & 2)) {
                    // ---- X3 epilogue: the fp32 result = activation (+ residual hi + residual lo), split into (hi, lo) bf16 and stored to
                    // both planes, round by round (a round's residual, both planes, is loaded one round ahead).  Every lane reads its own
                    // pixels' residual (both planes) before it stores them: in-place layers (the alpha residual into the view stack, the
                    // encoder's ResidualBlock) stay safe.
                    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): see the plain epilogue
                    int le = lane;
                    asm volatile("" : "+v"(le));
                    const int c15e = le & 15, qe = le >> 4;
                    // (wave-uniform, but hipcc may hold the output base in vector registers and then wraps every store in a waterfall loop)
                    const size_t ob_ = (size_t)outp;
                    unsigned char* const outu = (unsigned char*)((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ob_) |
                                                                 ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ob_ >> 32)) << 32));
                    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)outu, 0, (int)(hw * ROW), 0x00020000);
                    const __amdgpu_buffer_rsrc_t rs_out_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(outu + p.out_lo), 0, (int)(hw * ROW), 0x00020000);
                    const size_t res_lo_off = RESM == 2 ? p.stack_lo : p.res_lo;
                    lane_row_t rq[8][4];                                                      // [plane * 4 + round][j]
                    auto res_load = [&](int r, int pl) __attribute__((always_inline)) {      // plane pl of round r
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            rq[pl * 4 + r][j] = __builtin_nontemporal_load((const lane_row_t*)(res_src(r, j) + (pl ? res_lo_off : (size_t)0)));
                    };
                    // Residual schedule.  COUT = 64 (a lane's share of a row is 8 bytes): both planes of a round two rounds ahead.  COUT = 128
                    // (16 bytes: a round is 32 registers beside the 128 accumulators): round 0 and the hi plane of round 1 up front, and as
                    // the rounds retire their accumulators the prefetch deepens (lo 1 + round 2 behind round 0, round 3 behind round 1).
                    if (RES) {
                        res_load(0, 0); res_load(0, 1); res_load(1, 0);
                        if (LB == 8) res_load(1, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);               // (hipcc would otherwise start every round's loads here and spill them)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gy = y0 + 2 * w + (r >> 1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            lane_row_t oh, ol;
                            const lane_row_t rh = rq[r][j], rl = rq[4 + r][j];
#pragma unroll
                            for (int i = 0; i < NCB / 2; ++i) {
                                float xa = acc[2 * i][r][j], xb = acc[2 * i + 1][r][j];
                                xa = __builtin_amdgcn_fmed3f(xa, act_slope * xa, act_pick);
                                xb = __builtin_amdgcn_fmed3f(xb, act_slope * xb, act_pick);
                                if (RES) {
                                    const float ra = __uint_as_float(rh[i] << 16), rb = __uint_as_float(rh[i] & 0xffff0000u);
                                    if (RESM == 3) { xa = ra + res_alpha * xa; xb = rb + res_alpha * xb; }
                                    else { xa += ra; xb += rb; }
                                    xa += __uint_as_float(rl[i] << 16); xb += __uint_as_float(rl[i] & 0xffff0000u);
                                }
                                const unsigned h = pack2_bf16(xa, xb);
                                oh[i] = h;
                                ol[i] = pack2_bf16(xa - __uint_as_float(h << 16), xb - __uint_as_float(h & 0xffff0000u));
                            }
                            const int gx = x0 + 16 * (r & 1) + 4 * qe + j;
                            const unsigned voff = (unsigned)((gy * W + gx) * ROW + c15e * LB) | ((unsigned)(W - 1 - gx) & OOB6) | (gy < H ? 0u : OOB6);
                            if constexpr (LB == 16) {
                                __builtin_amdgcn_raw_buffer_store_b128(oh, rs_out, voff, 0, V6_ST_AUX);
                                __builtin_amdgcn_raw_buffer_store_b128(ol, rs_out_lo, voff, 0, V6_ST_AUX);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b64(oh, rs_out, voff, 0, V6_ST_AUX);
                                __builtin_amdgcn_raw_buffer_store_b64(ol, rs_out_lo, voff, 0, V6_ST_AUX);
                            }
                        }
                        if (RES) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (LB == 16) {
                                if (r == 0) { res_load(1, 1); res_load(2, 0); res_load(2, 1); }
                                if (r == 1) { res_load(3, 0); res_load(3, 1); }
                            } else if (r < 2) { res_load(r + 2, 0); res_load(r + 2, 1); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                } else
                if (tg == 2 && cc == NCC - 1 && !(V6_ABL & 2)) {
                    // ---- epilogue of this tile: registers and global memory only (accumulator layout: header)
                    // Nothing is in flight here (the counted wait above, halo_out == 0), but hipcc cannot see inline-asm waits: as long as it
                    // believes an LDS-DMA pending it answers the first use of a plain load with vmcnt(0), i.e. with a wait for the rounds
                    // issued behind it as well.  A wait it can see (free at this point) lets it count from here on.
                    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
                    int le = lane;
                    asm volatile("" : "+v"(le));                    // every lane-derived address below is formed here, per tile
                    const int c15e = le & 15, qe = le >> 4;
                    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, (int)(hw * ROW), 0x00020000);
                    lane_row_t rq[4][4];                                                      // [round][j]
                    auto res_load = [&](int r) __attribute__((always_inline)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (V6_ABL & 16) { rq[r][j] = lane_row_t{}; rq[r][j][0] = (unsigned)le; }
#if V6_RES_NT
                            else rq[r][j] = __builtin_nontemporal_load((const lane_row_t*)res_src(r, j));
#else
                            else rq[r][j] = *(const lane_row_t*)res_src(r, j);
#endif
                        }
                    };
                    // rounds 1 and 2 go into the (now dead) fragment registers, round 3 follows when round 0 is done
                    if (RES) {
                        if (FIFO) {
                            res_load(1); res_load(2);
#pragma unroll
                            for (int j = 0; j < 4; ++j) rq[0][j] = *(const lane_row_t*)(smem + GEO::OFF_FIFO + w * GEO::FIFO_WAVE + j * 1024 + le * 16);
                        } else { res_load(0); res_load(1); }
                    }
                    __builtin_amdgcn_sched_barrier(0);               // (hipcc would otherwise start all four rounds' loads here and spill them)
                    {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int gy = y0 + 2 * w + (r >> 1);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                lane_row_t o;
                                const lane_row_t rv = rq[r][j];
#pragma unroll
                                for (int i = 0; i < NCB / 2; ++i) {
                                    float xa = acc[2 * i][r][j], xb = acc[2 * i + 1][r][j];
                                    xa = __builtin_amdgcn_fmed3f(xa, act_slope * xa, act_pick);
                                    xb = __builtin_amdgcn_fmed3f(xb, act_slope * xb, act_pick);
                                    if (RES) {
                                        const float ra = __uint_as_float(rv[i] << 16), rb = __uint_as_float(rv[i] & 0xffff0000u);
                                        if (RESM == 3) { xa = ra + res_alpha * xa; xb = rb + res_alpha * xb; }
                                        else { xa += ra; xb += rb; }
                                    }
                                    o[i] = pack2_bf16(xa, xb);
                                }
                                const int gx = x0 + 16 * (r & 1) + 4 * qe + j;
                                // a pixel outside the image gets an offset the descriptor's range check drops: no branch around the store
                                const unsigned voff = (unsigned)((gy * W + gx) * ROW + c15e * LB) | ((unsigned)(W - 1 - gx) & OOB6) | (gy < H ? 0u : OOB6);
                                if (V6_ABL & 8) asm volatile("" :: "v"(o), "v"(voff));
                                else if constexpr (LB == 16) __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, voff, 0, V6_ST_AUX);
                                else __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, voff, 0, V6_ST_AUX);
                            }
                            if (RES && (FIFO ? r == 0 : r < 2)) res_load(FIFO ? 3 : r + 2);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
#ifdef V6_STAMP
                if (cc == NCC - 1 && tg == 2) V6_ST(16);
#endif
                barrier6();
                V6_ST(sb_ + 3);
#ifdef V6_STAMP
                stamp_on = so_;
#endif
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            stage(std::integral_constant<int, 2>{});
            if (X3) { if (++ps == 3) { ps = 0; ++c; } } else ++c;
        }
        cur_m = nxt_m; cur_t = nxt_t;
    }
    wait_vm6<0>();                                          // nothing of this workgroup may still be in flight when it ends
#ifdef V6_STAMP
    if (RESM == 2 && !PAIR && ntl >= 32 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 24; ++i) hrn_v6_stamps[(bid * 8 + w) * 24 + i] = st[i];
    }
#endif
}

template <int CIN, int COUT, int RESM, bool PAIR, bool X3>
int launch_v6(const ConvParams& p, long grid, hipStream_t stream) {
    typedef G6<COUT, X3> GEO;
    static_assert(GEO::LDS_BYTES <= 160 * 1024, "LDS budget");
    { const int rc_lds = hrn_allow_lds((const void*)conv3x3_v6_kernel<CIN, COUT, RESM, PAIR, X3>, GEO::LDS_BYTES); if (rc_lds) return rc_lds; }
    hipLaunchKernelGGL((conv3x3_v6_kernel<CIN, COUT, RESM, PAIR, X3>), dim3((unsigned)grid), dim3(512), GEO::LDS_BYTES, stream, p);
    HRN_LAUNCH_CHECK();
    return 0;
}

// tiles, the 32-bit arithmetic limits, the persistent grid: shared by both launchers.  Returns 0, or -100 when not applicable.
inline int v6_grid(const ConvParams& p, int cin, long& grid) {
    const long tiles = (long)((p.W + T6_W - 1) / T6_W) * ((p.H + T6_H - 1) / T6_H);
    const long total = tiles * p.M;
    HRN_CHECK(total > 0, -2, "conv3x3_v6: bad tile count %ld", total);
    if (total >= (1L << 30) || (long)p.H * p.W * (cin * 2) >= (1L << 31)) return -100;     // 32-bit tile / in-image byte arithmetic
    grid = hrn_device_cus();
    if (total < grid) grid = total;
    if (grid >= 8) grid &= ~7L;
    return 0;
}

}  // namespace
