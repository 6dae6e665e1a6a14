// conv3x3 128 -> 128, bf16, TWO TEAMS: the fusion ResidualBlock's layers (HRNet.py:90-94, :113-119) with the epilogue and every DMA
// issue taken out of the matrix pipe's time.
//
// conv3x3_v6 (conv3x3_v6_impl.h) runs eight MFMA waves in lock-step: all of them multiply, then all of them run the epilogue (activation,
// residual, rounding, stores) while the matrix pipe idles - 13-17 k of a tile's 67 k cycles (profiles/r02_final_v6_stamps.txt); with the
// epilogue compiled out the layer takes 0.60 instead of 0.74 ms (tools/v6_abl.sh 2).  Here the workgroup is two teams of four waves, one
// wave of each team on every SIMD, that alternate between
//   ON   the twelve stages (4 chunks of 32 input channels x 3 tap rows) of one 8 x 32-pixel tile: fragment reads and MFMAs only, the
//        v6 stage body; a wave owns two pixel rows x all 128 couts = 128 accumulator registers;
//   OFF  twelve segments, in step with the other team's stages: issue ALL the LDS-DMA the workgroup needs (the ON team's next weight
//        stage and halo chunk, and this team's own next tile's first chunk / first stage), and finish this team's previous tile from its
//        accumulators - residual by LDS-DMA into a 4 KB FIFO per wave (no vector-register load anywhere in the kernel: hipcc answers the
//        first use of one with vmcnt(0) while a DMA is in flight), activation, one bf16 rounding, whole-row stores.
// One workgroup barrier per stage / segment.  The matrix pipe sees one wave per SIMD at a time, with nothing but ds_reads between its
// MFMAs; the epilogue, the DMA issue cost (~100 cycles per instruction) and the store tail all run beside the other team's MFMAs.
// Cost: the weights are streamed once per 256-pixel tile instead of once per 512 (L2 -> LDS: ~12 B / clock / CU, all L2 hits).
// LDS: 2 x 24,576 (weight ring, used by the team that is ON) + 2 teams x 2 x 22,528 (halo chunks) + 512 (bias) + 4 x 4,096 (residual
// FIFO of the OFF team) = 156,160 B.
// Ordering rules as in v6: a wave waits for its own DMAs with a counted vmcnt BEFORE the barrier that precedes the stage reading them;
// a buffer is re-filled only after a barrier every reader of its previous contents has passed.  Within a segment the OFF wave issues, in
// this order, the residual DMA, the weight pieces, the halo pieces, then its stores: waiting until only the halo pieces (tap rows 0 and
// 1: they have until the chunk's last stage) and the stores of this segment are outstanding retires the weights and everything older.
#include <type_traits>
#include "conv3x3.h"

#ifndef V9_ON_W
#define V9_ON_W 2          // weight pieces (of a stage's 24) each ON wave issues from its MFMA gaps; the OFF waves issue 6 - V9_ON_W each
#endif
#ifndef V9_ST_AUX
#define V9_ST_AUX 2        // output stores non-temporal (as v6)
#endif

#ifdef V9_STAMP      // diagnostic build only (tools/stamps/read_v9.py): s_memtime stamps of phases 4 and 5 of the +res layer's largest launch
__device__ unsigned long long hrn_v9_stamps[256 * 8 * 80];
extern "C" int hrn_dbg_read_stamps_v9(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hrn_v9_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#define V9_ST(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (stamp_on) st[i] = t_; } while (0)
#else
#define V9_ST(i) do {} while (0)
#endif

namespace {

constexpr int T9_H = 8, T9_W = 32, HW9 = T9_W + 2;
constexpr int NPIX9 = (T9_H + 2) * HW9;                    // 340 halo pixels
constexpr int N_IN9 = (NPIX9 * 64 + 1023) / 1024;          // 22 DMA pieces of 1 KB per 32-channel halo chunk
constexpr int IN_BYTES9 = N_IN9 * 1024;                    // 22,528
constexpr unsigned OOB9 = 0x80000000u;

template <int COUT> struct G9 {
    static constexpr int NCB = COUT / 16, NQ = NCB / 2;
    static constexpr int TAP_BYTES = COUT * 64, WST = 3 * TAP_BYTES, W_PIECES = WST / 1024;     // 24,576 | 24
    static constexpr int OFF_IN = 2 * WST;                                                      // [team][buffer]
    static constexpr int OFF_BIAS = OFF_IN + 4 * IN_BYTES9;
    static constexpr int OFF_FIFO = OFF_BIAS + 512;                                             // 4 KB per wave of the OFF team
    static constexpr int ROW = COUT * 2, LB = NCB * 2;
    static constexpr int LDS_BYTES = OFF_FIFO + 4 * 4096;
};

typedef __attribute__((address_space(3))) void* lds_ptr9;
__device__ __forceinline__ int swz9(int row) { return ((row >> 2) & 1) << 1; }
__device__ __forceinline__ void wait_vm9_rt(int n) {       // n is wave-uniform
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void barrier9() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N, int I = 0, class F> __device__ __forceinline__ void static_for9(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for9<N, I + 1>(f); }
}

// RESM: 0 none | 2 the pair gather z (t2 = z + PReLU(conv(t1))).  PAIR: the conv input is the pair gather of the view stack.
template <int RESM, bool PAIR>
__global__ __launch_bounds__(512, 2) void conv3x3_v9_kernel(const ConvParams p) {
    constexpr int COUT = 128;
    typedef G9<COUT> GEO;
    constexpr int NCB = GEO::NCB, NQ = GEO::NQ, NSTEP = 3 * NQ, WST = GEO::WST, TAP_BYTES = GEO::TAP_BYTES;
    constexpr int OFF_IN = GEO::OFF_IN, ROW = GEO::ROW, LB = GEO::LB;
    constexpr bool RES = RESM != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bias_lds = (float*)(smem + GEO::OFF_BIAS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = w >> 2, tw = w & 3;
    const int c15 = lane & 15, q = lane >> 4;
    const int H = p.H, W = p.W;
    const unsigned hw = (unsigned)(H * W);
    const unsigned tiles_x = (W + T9_W - 1) / T9_W, tiles_y = (H + T9_H - 1) / T9_H;
    const unsigned tiles = tiles_x * tiles_y;
    const unsigned total = tiles * (unsigned)p.M;
    const unsigned G = gridDim.x, bid = blockIdx.x;
    const unsigned slot0 = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;      // each XCD walks a contiguous run of tiles
    if (slot0 >= total) return;
    const int ntl = (int)((total - slot0 + G - 1) / G);     // tiles of this workgroup: local tile l is done by team l & 1 in phase l
    constexpr unsigned in_pitch = PAIR ? 128u : 256u;
    const unsigned char* const src0 = (const unsigned char*)(PAIR ? p.stack : p.in);
    const unsigned img_bytes = hw * in_pitch;

    // local tile l -> (image m, tile t in the image)
    auto tile_of = [&](int l, unsigned& m, unsigned& t) __attribute__((always_inline)) {
        const unsigned s = slot0 + (unsigned)l * G;
        m = s / tiles; t = s - m * tiles;
    };
    // byte offsets of image m's input from src0: (view A, view B) of the pair gather (chunks 0-1 / 2-3), else one tensor image
    auto in_bases = [&](unsigned m, size_t& a, size_t& b) __attribute__((always_inline)) {
        if (PAIR) {
            const unsigned bb = m / (unsigned)p.pair_h, i = m - bb * (unsigned)p.pair_h;
            a = ((size_t)bb * p.pair_vs + i) * hw * 128;
            b = ((size_t)bb * p.pair_vs + (p.pair_last - i)) * hw * 128;
        } else {
            a = b = (size_t)m * hw * 256;
        }
    };
    // per-lane byte offsets of this wave's halo pieces (j = tw + 4 jj < 22) of tile t: lane i -> halo pixel 16 j + (i >> 2), physical
    // 16-byte chunk i & 3 = logical chunk ^ swz9(pixel); pixels outside the image (or beyond the halo) -> OOB9 (the descriptor writes zeros)
    auto tile_offsets = [&](unsigned t, unsigned (&ho)[6]) __attribute__((always_inline)) {
        const int ty = t / tiles_x;
        const int y0 = ty * T9_H, x0 = (t - ty * tiles_x) * T9_W;
        int lq = lane;
        asm volatile("" : "+v"(lq));
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            const int pix = (tw + 4 * jj) * 16 + (lq >> 2);
            const int lc = (lq & 3) ^ swz9(pix);
            const int py = pix / HW9, px = pix - py * HW9;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool ok = pix < NPIX9 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            ho[jj] = ok ? (unsigned)(gy * W + gx) * in_pitch + (unsigned)(lc * 16) : OOB9;
        }
    };
    // one halo piece of chunk c -> halo buffer `buf` of team `tm`
    auto dma_halo = [&](__amdgpu_buffer_rsrc_t rs, int c, int tm, int buf, int jj, unsigned voff) __attribute__((always_inline)) {
        const int j = tw + 4 * jj;
        if (j < N_IN9) {
            const unsigned soff = PAIR ? (unsigned)((c & 1) * 64) : (unsigned)(c * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr9)(smem + OFF_IN + (tm * 2 + buf) * IN_BYTES9 + j * 1024), 16, voff, soff, 0, 0);
        }
    };
    // one weight piece of stage (c, tg): piece qq = (tap column kx = qq / NCB, cout block jb = qq % NCB); row r of the block holds cout
    // NCB r + jb (the interleave that makes a lane's accumulators a contiguous piece of its pixels' rows: conv3x3_v6_impl.h)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, 9 * 128 * COUT * 2, 0x00020000);
    const unsigned w_lane_off = (unsigned)((lane >> 2) * (NCB * 128) + (((lane & 3) ^ swz9(lane >> 2)) << 4));
    auto dma_w_piece = [&](int c, int tg, int slot_, int qq) __attribute__((always_inline)) {
        const int kx = qq / NCB, jb = qq - kx * NCB;
        const unsigned soff = (unsigned)(((c >> 1) * 9 + tg * 3 + kx) * (COUT * 128) + (c & 1) * 64 + jb * 128);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr9)(smem + slot_ * WST + kx * TAP_BYTES + jb * 1024), 16, w_lane_off, soff, 0, 0);
    };

    const float act_slope = p.slope ? p.slope[0] : 1.f;                        // PReLU as one v_med3 (v6)
    const float act_pick = act_slope <= 1.f ? __builtin_inff() : -__builtin_inff();

    // fragment addresses (v6): A (weights) a_off + slot*WST + kx*TAP + cb*1024; B (pixels), one register per (halo row 2 tw + j, tap
    // column kx), for buffer 0 of this team; the second half of a pixel block lies 1,024 bytes further on
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned a_off = lds0 + (unsigned)(c15 * 64 + ((q ^ swz9(c15)) << 4));
    unsigned baddr[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int pix = (2 * tw + j) * HW9 + c15 + kx;
            baddr[j][kx] = lds0 + (unsigned)(OFF_IN + team * 2 * IN_BYTES9) + (unsigned)(pix << 6) + ((unsigned)(q << 4) ^ (unsigned)((pix & 4) << 3));
        }

    f32x4 acc[NCB][4];                                      // [cout block of 16][pixel block of 16]
#ifdef V9_STAMP
    unsigned long long st[80];
#pragma unroll
    for (int i = 0; i < 80; ++i) st[i] = 0;
    bool stamp_on = false;
#endif

    // ---- prologue: tile 0's first halo chunk (team 0, buffer 0) and first weight stage, by all eight waves
    if (tid < COUT) bias_lds[tid] = p.bias[tid];
    {
        unsigned m0, t0;
        tile_of(0, m0, t0);
        size_t a0, b0;
        in_bases(m0, a0, b0);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src0 + a0), 0, (int)img_bytes, 0x00020000);
        // pieces j = w + 8 jj
        int lq = lane;
        asm volatile("" : "+v"(lq));
        const int ty = t0 / tiles_x, y0 = ty * T9_H, x0 = (t0 - ty * tiles_x) * T9_W;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int j = w + 8 * jj;
            if (j < N_IN9) {
                const int pix = j * 16 + (lq >> 2);
                const int lc = (lq & 3) ^ swz9(pix);
                const int py = pix / HW9, px = pix - py * HW9;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool ok = pix < NPIX9 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                const unsigned voff = ok ? (unsigned)(gy * W + gx) * in_pitch + (unsigned)(lc * 16) : OOB9;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr9)(smem + OFF_IN + j * 1024), 16, voff, 0u, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dma_w_piece(0, 0, 0, w + 8 * k);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier9();

    // state of the OFF role: the tile whose epilogue this team runs, and the DMA offsets of the tiles it fetches for
    unsigned ho_on[6], ho_nx[6];
    size_t onA = 0, onB = 0, nxA = 0;
    int e_y0 = 0, e_x0 = 0;
    const unsigned char *resA = nullptr, *resB = nullptr;
    unsigned char* outp = nullptr;

    // the OFF role's state for phase `po` (this team is ON in phase po - 1): part 0 = the ON tile po, 1 = the tile after it, 2 = the
    // epilogue of this team's own tile po - 1
    auto prepare_off = [&](int po, int part) __attribute__((always_inline)) {
        if (part == 0) {
            if (po < ntl) { unsigned m, t; tile_of(po, m, t); in_bases(m, onA, onB); tile_offsets(t, ho_on); }
        } else if (part == 1) {
            if (po + 1 < ntl) { unsigned m, t; size_t dummy; tile_of(po + 1, m, t); in_bases(m, nxA, dummy); tile_offsets(t, ho_nx); }
        } else if (po >= 1) {
            unsigned m, t;
            tile_of(po - 1, m, t);
            const int ty_ = t / tiles_x;
            e_y0 = ty_ * T9_H; e_x0 = (t - ty_ * tiles_x) * T9_W;
            if (RESM == 2) {
                const unsigned bb = m / (unsigned)p.pair_h, i = m - bb * (unsigned)p.pair_h;
                resA = (const unsigned char*)p.stack + ((size_t)bb * p.pair_vs + i) * hw * 128;
                resB = (const unsigned char*)p.stack + ((size_t)bb * p.pair_vs + (p.pair_last - i)) * hw * 128;
            }
            outp = (unsigned char*)p.out + (size_t)m * hw * ROW;
        }
    };
    if (team == 1) { prepare_off(0, 0); prepare_off(0, 1); }         // team 1 starts in the OFF role

    u32x4 rq[4] = {};                                       // the residual of the epilogue round in progress (from the FIFO)
    for (int ph = 0; ph <= ntl; ++ph) {
        const bool have_on = ph < ntl, have_next = ph + 1 < ntl;
        const bool my_on = have_on && team == (ph & 1);
        const bool my_off = team != (ph & 1);
        const bool have_epi = my_off && ph >= 1;            // this team multiplied local tile ph - 1 in the previous phase
#ifdef V9_STAMP
        stamp_on = RESM == 2 && !PAIR && ntl >= 32 && (ph == 4 || ph == 5);
        const int sbase = (ph & 1) * 36;                    // per segment: start, work done, vm wait done (= barrier entered); + phase end
#endif
        if (my_on) {
            // accumulators start at the bias: every element of acc[cb][.] of this lane is channel NCB * c15 + cb
#pragma unroll
            for (int cb4 = 0; cb4 < NCB; cb4 += 4) {
                const f32x4 b = *(const f32x4*)(bias_lds + NCB * c15 + cb4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int pxb = 0; pxb < 4; ++pxb) acc[cb4 + e][pxb] = f32x4{b[e], b[e], b[e], b[e]};
            }
        }
        // (what the OFF role needs - the DMA offsets of the tiles it fetches for, the geometry of the tile it finishes - was computed at
        // the end of this team's previous ON stages, where the wave would otherwise idle at the barrier: prepare_off() below)
        // the residual of round r (= pixel block r of the wave), piece j: lane (q, c15) fetches its own share of pixel 4q + j of
        // z = cat(view i, partner): the 16 bytes that hold channels 8 c15 .. 8 c15 + 7
        auto res_src = [&](int r, int j) __attribute__((always_inline)) -> const unsigned char* {
            int lq = lane;
            asm volatile("" : "+v"(lq));
            const int c15r = lq & 15, qr = lq >> 4;
            const int gy = e_y0 + 2 * tw + (r >> 1), gyc = gy < H ? gy : H - 1;
            const int gx = e_x0 + 16 * (r & 1) + 4 * qr + j, gxc = gx < W ? gx : W - 1;
            const unsigned char* view = c15r >= 8 ? resB : resA;
            return view + ((unsigned)((gyc * W + gxc) * 128) + (unsigned)((c15r & 7) * 16));
        };

        for (int c = 0; c < 4; ++c) {
            // the ON team's chunk c sits in its buffer c & 1: move the B addresses over (chunk 0 of every tile is in buffer 0)
            if (my_on && c > 0) {
                const unsigned d = (c & 1) ? (unsigned)IN_BYTES9 : (unsigned)-IN_BYTES9;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) baddr[j][kx] += d;
            }
            auto stage = [&](auto tg_c) __attribute__((always_inline)) {
                constexpr int tg = decltype(tg_c)::value;
                const int s = 3 * c + tg;
                const int slot_r = s & 1;
                // the stage after this one: stage s + 1 of the ON tile, or stage 0 of the next tile
                const bool w_next = s < 11 ? have_on : have_next;
                const int c2w = s < 11 ? (tg == 2 ? c + 1 : c) : 0, tg2w = (tg + 1) % 3;
                V9_ST(sbase + 3 * s);
                if (my_on) {
                    // ---- NSTEP steps = 3 taps x NQ cout pairs, 8 MFMAs each: v6's hand-issued fragment stream with counted waits
                    const unsigned abase = a_off + (unsigned)(slot_r * WST);
                    bf16x8 fa[2][2], fb[2][4];
                    auto rd = [&](bf16x8& dst, unsigned addr, int imm) __attribute__((always_inline)) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm));
                    };
                    auto load_b1 = [&](int tap, int pxb) __attribute__((always_inline)) {
                        rd(fb[tap & 1][pxb], baddr[(pxb >> 1) + tg][tap], (pxb & 1) * 1024);
                    };
                    auto load_a1 = [&](int i, int k) __attribute__((always_inline)) {
                        rd(fa[i & 1][k], abase, (i / NQ) * TAP_BYTES + (i % NQ) * 2048 + k * 1024);
                    };
#pragma unroll
                    for (int pxb = 0; pxb < 4; ++pxb) load_b1(0, pxb);
                    load_a1(0, 0);
                    load_a1(0, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    static_for9<NSTEP>([&](auto i_c) __attribute__((always_inline)) {
                        constexpr int i = decltype(i_c)::value;
                        constexpr int qt = i % NQ, tap = i / NQ, bs = tap & 1;
                        constexpr bool a_next = i + 1 < NSTEP;
                        constexpr bool b_cur = qt == 1 && tap + 1 < 3;
                        constexpr bool b_prev = i >= 1 && ((i - 1) % NQ) == 1 && (i - 1) / NQ + 1 < 3;
                        constexpr int n0 = 1 + (b_prev ? 4 : 0);
                        constexpr int n4 = (b_prev ? 4 : 0) + (a_next ? 2 : 0) + (b_cur ? 2 : 0);
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            const int k = g >> 2, pxb = g & 3;
                            if (g == 0) {
                                if (n0 == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa[i & 1][0]), "+v"(fb[bs][0]));
                                else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa[i & 1][0]), "+v"(fb[bs][0]));
                            } else if (g == 4) {
                                if (n4 == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[i & 1][1]));
                                else if (n4 == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[i & 1][1]));
                                else if (n4 == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[i & 1][1]));
                                else if (n4 == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fa[i & 1][1]));
                                else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[i & 1][1]));
                            } else if (k == 0) asm volatile("" : "+v"(fb[bs][pxb]));
                            acc[qt * 2 + k][pxb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[bs][pxb], fa[i & 1][k], acc[qt * 2 + k][pxb], 0, 0, 0);
                            if (g < 2 && a_next) load_a1(i + 1, g);
                            if (g >= 2 && g < 6 && b_cur) load_b1(tap + 1, g - 2);
                            // two of the next stage's 24 weight pieces are issued from here (~60 cycles each among MFMAs: the OFF waves,
                            // which carry the other 16, the halo, the residual and the epilogue, are the longer side of a segment otherwise)
                            if (g == 7 && i < V9_ON_W && w_next) dma_w_piece(c2w, tg2w, slot_r ^ 1, 4 * (6 - V9_ON_W) + tw + 4 * i);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    if (c == 0) prepare_off(ph + 1, tg);    // (the wave would wait at the barrier for the OFF team otherwise)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // its two weight pieces, issued a stage ago in wave time
                } else if (my_off) {
                    int n_after = 0;                        // VMEM instructions this wave may leave in flight at the segment's barrier
                    // ---- (1) the residual of this round (DMA'd into the FIFO in the previous segment, the youngest of its operations)
                    if (RES && have_epi && tg == 1) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        int le = lane;
                        asm volatile("" : "+v"(le));
#pragma unroll
                        for (int j = 0; j < 4; ++j) rq[j] = *(const u32x4*)(smem + GEO::OFF_FIFO + tw * 4096 + j * 1024 + le * 16);
                    }
                    // ---- (2) the weights of the stage after this one: 16 of the 24 pieces (the ON waves issue the other 8)
                    if (w_next) {
#pragma unroll
                        for (int k = 0; k < 6 - V9_ON_W; ++k) dma_w_piece(c2w, tg2w, slot_r ^ 1, tw + 4 * k);
                    }
                    // ---- (3) halo: under chunk c of the ON tile its chunk c + 1 (c < 3), under its last chunk the first chunk of the next
                    // tile (which is this team's own): pieces 0-2 of the wave in tap row 0, 3-5 in tap row 1
                    if (tg < 2) {
                        if (c < 3) {
                            if (have_on) {
                                const int cn = c + 1;
                                const size_t hb = (PAIR && cn >= 2) ? onB : onA;
                                const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)(src0 + hb), 0, (int)img_bytes, 0x00020000);
#pragma unroll
                                for (int k = 0; k < 3; ++k) {
                                    const int jj = 3 * tg + k;
                                    dma_halo(rs_h, cn, ph & 1, cn & 1, jj, ho_on[jj]);
                                    if (tw + 4 * jj < N_IN9) ++n_after;
                                }
                            }
                        } else if (have_next) {
                            const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)(src0 + nxA), 0, (int)img_bytes, 0x00020000);
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const int jj = 3 * tg + k;
                                dma_halo(rs_h, 0, (ph + 1) & 1, 0, jj, ho_nx[jj]);
                                if (tw + 4 * jj < N_IN9) ++n_after;
                            }
                        }
                    }
                    // ---- (3b) residual of epilogue round r = c -> this wave's FIFO, read at the start of the next segment
                    if (RES && have_epi && tg == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)res_src(c, j),
                                                             (lds_ptr9)(smem + GEO::OFF_FIFO + tw * 4096 + j * 1024), 16, 0, 2);
                        n_after += 4;
                    }
                    // ---- (4) epilogue of this team's previous tile, round r = c: pixels j = 0, 1 in tap row 1's segment, 2, 3 in tap row 2's
                    if (have_epi && tg >= 1) {
                        int le = lane;
                        asm volatile("" : "+v"(le));
                        const int c15e = le & 15, qe = le >> 4;
                        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, (int)(hw * ROW), 0x00020000);
                        const int gy = e_y0 + 2 * tw + (c >> 1);
                        auto round = [&](auto r_c) __attribute__((always_inline)) {
                            constexpr int r = decltype(r_c)::value;
#pragma unroll
                            for (int jh = 0; jh < 2; ++jh) {
                                const int j = 2 * (tg - 1) + jh;
                                u32x4 o;
                                const u32x4 rv = rq[j];
#pragma unroll
                                for (int i = 0; i < NCB / 2; ++i) {
                                    float xa = acc[2 * i][r][j], xb = acc[2 * i + 1][r][j];
                                    xa = __builtin_amdgcn_fmed3f(xa, act_slope * xa, act_pick);
                                    xb = __builtin_amdgcn_fmed3f(xb, act_slope * xb, act_pick);
                                    if (RES) { xa += __uint_as_float(rv[i] << 16); xb += __uint_as_float(rv[i] & 0xffff0000u); }
                                    o[i] = pack2_bf16(xa, xb);
                                }
                                const int gx = e_x0 + 16 * (r & 1) + 4 * qe + j;
                                const unsigned voff = (unsigned)((gy * W + gx) * ROW + c15e * LB) | ((unsigned)(W - 1 - gx) & OOB9) | (gy < H ? 0u : OOB9);
                                __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, voff, 0, V9_ST_AUX);
                            }
                        };
                        if (c == 0) round(std::integral_constant<int, 0>{});
                        else if (c == 1) round(std::integral_constant<int, 1>{});
                        else if (c == 2) round(std::integral_constant<int, 2>{});
                        else round(std::integral_constant<int, 3>{});
                        n_after += 2;
                    }
                    // the weights (and everything older: the residual DMA, earlier halo pieces and stores) have landed once only this segment's
                    // halo pieces and stores are outstanding
                    V9_ST(sbase + 3 * s + 1);
                    wait_vm9_rt(n_after);
                }
                V9_ST(sbase + 3 * s + 2);
                barrier9();
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            stage(std::integral_constant<int, 2>{});
        }
        // the ON team leaves its B addresses on buffer 1 (chunk 3): back to buffer 0 for its next tile
        if (my_on) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) baddr[j][kx] -= (unsigned)IN_BYTES9;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // nothing of this workgroup may still be in flight when it ends
#ifdef V9_STAMP
    if (RESM == 2 && !PAIR && ntl >= 32 && lane == 0) {
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
        st[79] = t_;
#pragma unroll
        for (int i = 0; i < 80; ++i) hrn_v9_stamps[(bid * 8 + w) * 80 + i] = st[i];
    }
#endif
}

template <int RESM, bool PAIR>
int launch_v9(const ConvParams& p, long grid, hipStream_t stream) {
    typedef G9<128> GEO;
    static_assert(GEO::LDS_BYTES <= 160 * 1024, "LDS budget");
    { const int rc_lds = hrn_allow_lds((const void*)conv3x3_v9_kernel<RESM, PAIR>, GEO::LDS_BYTES); if (rc_lds) return rc_lds; }
    hipLaunchKernelGGL((conv3x3_v9_kernel<RESM, PAIR>), dim3((unsigned)grid), dim3(512), GEO::LDS_BYTES, stream, p);
    HRN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// bf16 128 -> 128: residual none or the pair gather (res_mode 2), input plain or the pair gather.  Returns -100 when not applicable.
int hrn_launch_conv3x3_v9(int cout, const ConvParams& p, hipStream_t stream) {
    if (p.scale || p.relu || cout != 128) return -100;
    if (p.res_mode != 0 && p.res_mode != 2) return -100;
    if ((p.in_pair || p.res_mode == 2) && p.pair_h <= 0) return -100;
    if (p.out_h > 0) return -100;
    const long tiles = (long)((p.W + T9_W - 1) / T9_W) * ((p.H + T9_H - 1) / T9_H);
    const long total = tiles * p.M;
    HRN_CHECK(total > 0, -2, "conv3x3_v9: bad tile count %ld", total);
    if (total >= (1L << 30) || (long)p.H * p.W * 256 >= (1L << 31)) return -100;     // 32-bit tile / in-image byte arithmetic
    long grid = hrn_device_cus();
    if (total < grid) grid = total;
    if (grid >= 8) grid &= ~7L;
    const double px = (double)p.M * p.H * p.W;
    HrnProfScope prof(p.res_mode ? "conv3x3_bf16_128x128+res" : "conv3x3_bf16_128x128", 2.0 * 128 * 128 * 9 * px,
                      px * 2 * (128 + 128 + (p.res_mode ? 128 : 0)), stream);
    if (p.in_pair) return p.res_mode ? launch_v9<2, true>(p, grid, stream) : launch_v9<0, true>(p, grid, stream);
    return p.res_mode ? launch_v9<2, false>(p, grid, stream) : launch_v9<0, false>(p, grid, stream);
}
