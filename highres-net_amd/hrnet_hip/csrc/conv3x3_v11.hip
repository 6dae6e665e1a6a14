// conv3x3_v11: the bf16 128 -> {128, 64} convolutions of a fusion level (HRNet.py:17-22,93-97,114-131) with the epilogue interleaved
// INTO each wave's own MFMA stream.
//
// conv3x3_v6 (same tile, same per-wave work: DESIGN.md 3.1) spends 19 % of a launch in an epilogue during which the matrix pipe idles.
// Putting another wave's MFMAs beside it failed twice in round 3 (two teams on two tiles: conv3x3_v9, doubled weight stream; the
// wave groups one stage apart: conv3x3_v10 - the epilogue of ONE wave is a 7-8 k-cycle chain of residual latency, VALU and stores, and in
// lock step the partner has a single stage to put beside it).  So the chain is cut up and laid into the gaps of the wave's own MFMAs:
//
//   * the LAST stage of a tile (chunk 3, tap row 2) and the FIRST stage of the next one run their MFMAs ordered by PIXEL ROW of the
//     wave (pixel blocks 0, 1 = row 2w, then 2, 3 = row 2w + 1) instead of tap by tap over all four blocks.  After the first half of the
//     last stage the accumulators of row 2w are final: their epilogue (rounds 0, 1) is issued, a few instructions at a time, behind the
//     MFMAs of the second half; the epilogue of row 2w + 1 (rounds 2, 3) goes behind the first half of the NEXT tile's first stage,
//     which only touches the (re-initialised) accumulators of row 2w.  No second set of accumulators, no LDS.
//   * the residual of rounds 0, 1 is requested at the start of the last stage, that of rounds 2, 3 at its end: half a stage or more
//     before the first use, into registers the halved B fragment set leaves free.
//   * cost: the A (weight) fragments of those two stages are read twice (once per pixel row): 2 of 12 stages.
//
// Everything else is conv3x3_v6's: 512-pixel tiles, one persistent 512-thread workgroup per CU on an XCD-contiguous run of tiles,
// descriptor-based LDS-DMA for halo and weights from the MFMA gaps, 2-slot weight ring, double-buffered halo chunk, hand-issued
// fragment reads with counted waits, stores of whole pixel rows straight from the accumulators.  From conv3x3_v10: the halo swizzle by
// pixel COLUMN, which makes a tap row a pure byte offset and lets ONE stage body serve the ten ordinary stages (v6 unrolls three).
#include <type_traits>
#include "conv3x3.h"

#ifndef V11_ABL
#define V11_ABL 0          // timing-only ablations (results are WRONG when set): 2 = no output stores | 4 = no epilogue at all
#endif

namespace {

constexpr int TH = 16, TW = 32;
constexpr int HWID = TW + 2;                                // halo width 34
constexpr int NPIX = (TH + 2) * HWID;                       // 612 halo pixels
constexpr int N_IN = (NPIX * 64 + 1023) / 1024;             // 39 DMA pieces of 1 KB per 32-channel halo chunk
constexpr int IN_BYTES = N_IN * 1024;
constexpr int ROWB = HWID * 64;                             // one halo row of a chunk: 2,176 bytes
constexpr unsigned OOB = 0x80000000u;                       // byte offset no descriptor of this kernel covers
constexpr int NCH = 4;                                      // chunks of 32 input channels (CIN = 128)

template <int COUT> struct G11 {
    static constexpr int NCB = COUT / 16;                   // cout blocks of 16 per wave
    static constexpr int NQ = NCB / 2;                      // steps per tap (2 cout blocks x 4 pixel blocks = 8 MFMAs each)
    static constexpr int TAP_BYTES = COUT * 64;             // one tap x 32 cin
    static constexpr int WST = 3 * TAP_BYTES;               // one stage: 24,576 | 12,288
    static constexpr int W_PIECES = WST / 1024;             // 24 | 12
    static constexpr int OFF_IN = 2 * WST;                  // behind the two-slot weight ring
    static constexpr int OFF_BIAS = OFF_IN + 2 * IN_BYTES;
    static constexpr int ROW = COUT * 2;                    // bytes per output pixel
    static constexpr int LB = NCB * 2;                      // bytes of a pixel's row one lane holds
    static constexpr int LDS_BYTES = OFF_BIAS + 512;
};

typedef __attribute__((address_space(3))) void* lds_ptr;

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm_rt(int n) {         // n is wave-uniform, 0..5
    switch (n) {
        case 1: wait_vm<1>(); break;
        case 2: wait_vm<2>(); break;
        case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break;
        case 5: wait_vm<5>(); break;
        default: wait_vm<0>(); break;
    }
}
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ int swz_w(int row) { return ((row >> 2) & 1) << 1; }       // weight blocks: by row of the block
__device__ __forceinline__ int swz_c(int col) { return ((col >> 2) & 1) << 1; }       // halo: by pixel column

template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<N, I + 1>(f); }
}

template <int COUT, int RESM, bool PAIR>
__global__ __launch_bounds__(512, 2) void conv3x3_v11_kernel(const ConvParams p) {
    typedef G11<COUT> GEO;
    constexpr int NCB = GEO::NCB, NQ = GEO::NQ, NSTEP = 3 * NQ, WST = GEO::WST, TAP_BYTES = GEO::TAP_BYTES;
    constexpr int OFF_IN = GEO::OFF_IN, ROW = GEO::ROW, LB = GEO::LB;
    typedef typename std::conditional<LB == 16, u32x4, u32x2>::type lane_row_t;       // a lane's share of one pixel's row
    constexpr bool RES = RESM != 0;
    constexpr int BAL = COUT == 128 ? 4 : 0;                // eighths of a stage during which waves 4-7 run at raised priority (v6's V6_BAL)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bias_lds = (float*)(smem + GEO::OFF_BIAS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c15 = lane & 15, q = lane >> 4;
    const int H = p.H, W = p.W;
    const unsigned hw = (unsigned)(H * W);
    const unsigned tiles_x = (W + TW - 1) / TW;
    const unsigned tiles_y = (H + TH - 1) / TH;
    const unsigned tiles = tiles_x * tiles_y;
    const unsigned total = tiles * (unsigned)p.M;
    const unsigned G = gridDim.x;
    const unsigned bid = blockIdx.x;
    const unsigned slot0 = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;      // each XCD walks a contiguous run of tiles
    if (slot0 >= total) return;
    const int ntl = (int)((total - slot0 + G - 1) / G);
    const unsigned step_m = G / tiles, step_t = G - step_m * tiles;
    constexpr unsigned in_pitch = PAIR ? 128u : 256u;
    const unsigned char* const src0 = (const unsigned char*)(PAIR ? p.stack : p.in);
    const unsigned img_bytes = hw * in_pitch;               // < 2^31 (checked by the launcher)

    auto next_tile = [&](unsigned& m, unsigned& t) __attribute__((always_inline)) {
        t += step_t; m += step_m;
        if (t >= tiles) { t -= tiles; ++m; }
    };
    // where image m of the input lives, as byte offsets from src0: (view A, view B) for the pair gather (chunks 0-1 / 2-3), else one image
    auto in_bases = [&](unsigned m, size_t& a, size_t& b) __attribute__((always_inline)) {
        if (PAIR) {
            const unsigned bb = m / (unsigned)p.pair_h, i = m - bb * (unsigned)p.pair_h;
            a = ((size_t)bb * p.pair_vs + i) * hw * 128;
            b = ((size_t)bb * p.pair_vs + (p.pair_last - i)) * hw * 128;
        } else {
            a = b = (size_t)m * hw * in_pitch;
        }
    };
    // per-lane byte offsets of this wave's halo pieces for tile t (pieces j = w + 8 jj < 39; lane i -> halo pixel j*16 + (i >> 2),
    // physical 16-byte chunk i & 3 = logical chunk ^ swz_c(column)); invalid pixels -> OOB: the DMA writes zeros for them
    unsigned hoff[5];
    auto tile_offsets = [&](unsigned t) __attribute__((always_inline)) {
        const int ty = t / tiles_x;
        const int y0 = ty * TH, x0 = (t - ty * tiles_x) * TW;
        int lq = lane;
        asm volatile("" : "+v"(lq));                        // keep the per-piece geometry out of long-lived registers
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int pix = (w + 8 * jj) * 16 + (lq >> 2);
            const int py = pix / HWID, px = pix - py * HWID;
            const int lc = (lq & 3) ^ swz_c(px);
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool ok = pix < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            hoff[jj] = ok ? (unsigned)(gy * W + gx) * in_pitch + (unsigned)(lc * 16) : OOB;
        }
    };
    // one halo piece: chunk c (32 channels = 64 bytes of a pixel) of the image behind `rs` -> input buffer `buf`
    auto dma_halo = [&](__amdgpu_buffer_rsrc_t rs, int c, int buf, int jj, unsigned voff) __attribute__((always_inline)) {
        const int j = w + 8 * jj;
        if (j < N_IN) {
            const unsigned soff = PAIR ? (unsigned)((c & 1) * 64) : (unsigned)(c * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + OFF_IN + buf * IN_BYTES + j * 1024), 16, voff, soff, 0, 0);
        }
    };
    // one weight piece of stage (c, tg): piece qq = (tap kx = qq / NCB, cout block jb = qq % NCB): 16 couts x 64 bytes; lane i -> row
    // i >> 2 of the block = cout NCB * (i >> 2) + jb (the interleave that makes a lane's accumulators a contiguous piece of its pixels'
    // rows: conv3x3_v6.hip), physical chunk i & 3 = logical (i & 3) ^ swz_w(i >> 2)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (int)(9u * 128 * COUT * 2u), 0x00020000);
    const unsigned w_lane_off = (unsigned)((lane >> 2) * (NCB * 128) + (((lane & 3) ^ swz_w(lane >> 2)) << 4));
    auto dma_w = [&](int c, int tg, int slot_, int t3) __attribute__((always_inline)) {
        const int qq = w + 8 * t3;
        if (qq < GEO::W_PIECES) {
            const int kx = qq / NCB, jb = qq - kx * NCB;
            const unsigned soff = (unsigned)(((c >> 1) * 9 + tg * 3 + kx) * (COUT * 128) + (c & 1) * 64 + jb * 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(smem + slot_ * WST + kx * TAP_BYTES + jb * 1024), 16, w_lane_off, soff, 0, 0);
        }
    };
    constexpr int NW3 = (GEO::W_PIECES + 7) / 8;            // weight pieces a wave issues per stage: 3 | 2
    constexpr int N_ITEMS = NW3 + 3;                        // DMA items of a stage: the next stage's weights, then up to 3 pieces of the next halo chunk
    static_assert(N_ITEMS <= NSTEP, "one DMA item per MFMA step");
    const int n_in = w < (N_IN & 7) ? (N_IN >> 3) + 1 : (N_IN >> 3);      // halo pieces of this wave per chunk: 5 (wave 7: 4)

    // PReLU(x) = median(x, s x, +inf | -inf): one instruction for every slope (conv3x3_v6.hip); no activation == slope 1
    const float act_slope = p.slope ? p.slope[0] : 1.f;
    const float act_pick = act_slope <= 1.f ? __builtin_inff() : -__builtin_inff();

    // fragment addresses.  A, cout block cb: a_off + slot*WST + kx*TAP + cb*1024.  B, pixel block pxb = (row pxb >> 1, column half
    // pxb & 1) of tap (row tg, column kx): halo pixel (2w + (pxb >> 1) + tg) * 34 + c15 + kx (+ 16): one register per (row 2w + j,
    // j = 0 | 1, column kx) for tap row 0 of buffer 0; the stage adds (tap row) * ROWB + (buffer) * IN_BYTES
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned a_off = lds0 + (unsigned)(c15 * 64 + ((q ^ swz_w(c15)) << 4));
    unsigned baddr0[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int col = c15 + kx;
            baddr0[j][kx] = lds0 + (unsigned)OFF_IN + (unsigned)(((2 * w + j) * HWID + col) << 6) + ((unsigned)((q ^ swz_c(col)) << 4));
        }

    f32x4 acc[NCB][4];                                      // [cout block of 16][pixel block of 16]


    // ---- the tile being computed, and the geometry of the tile whose accumulators are being stored (the same, except during the first
    // half of a tile's first stage, when rounds 2, 3 of the previous tile are still on their way out)
    unsigned cur_m = slot0 / tiles, cur_t = slot0 - cur_m * tiles;
    size_t inA, inB;
    in_bases(cur_m, inA, inB);
    tile_offsets(cur_t);
    int y0 = 0, x0 = 0;
    const unsigned char *resA = nullptr, *resB = nullptr;
    unsigned char* outp = nullptr;
    float res_alpha = 1.f;
    auto geometry = [&](unsigned m, unsigned t) __attribute__((always_inline)) {
        const int ty_ = t / tiles_x;
        y0 = ty_ * TH; x0 = (t - ty_ * tiles_x) * TW;
        size_t oimg = m;
        res_alpha = 1.f;
        if (p.out_h > 0) {
            const unsigned ob = m / (unsigned)p.out_h, oi = m - ob * (unsigned)p.out_h;
            oimg = (size_t)ob * p.out_vs + oi;
            if (RESM == 3) {
                resA = resB = (const unsigned char*)p.res + ((size_t)ob * p.res_vs + oi) * hw * 128;
                if (p.alphas) res_alpha = p.alphas[(size_t)ob * p.alpha_vs + (p.pair_last - oi)];
            }
        }
        if (RESM == 2) {
            const unsigned bb = m / (unsigned)p.pair_h, i = m - bb * (unsigned)p.pair_h;
            resA = (const unsigned char*)p.stack + ((size_t)bb * p.pair_vs + i) * hw * 128;
            resB = (const unsigned char*)p.stack + ((size_t)bb * p.pair_vs + (p.pair_last - i)) * hw * 128;
        }
        outp = (unsigned char*)p.out + oimg * hw * ROW;
    };
    auto init_acc = [&](auto lo_c, auto hi_c) __attribute__((always_inline)) {      // accumulators of pixel blocks [lo, hi) start at the bias
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
#pragma unroll
        for (int cb4 = 0; cb4 < NCB; cb4 += 4) {
            const f32x4 b = *(const f32x4*)(bias_lds + NCB * c15 + cb4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int pxb = lo; pxb < hi; ++pxb) acc[cb4 + e][pxb] = f32x4{b[e], b[e], b[e], b[e]};
        }
    };

    // ---- the epilogue, in sub-pieces.  acc[cb][pxb][e] of lane (q, c15) is pixel 4q + e of pixel block pxb, channel NCB c15 + cb: the
    // NCB values a lane holds of one pixel are LB contiguous bytes of its row, sixteen lanes the whole row (conv3x3_v6.hip).  Round r =
    // pixel block r; piece (r, j) = pixel 4q + j of it; a piece is SUBS sub-pieces of PPS channel pairs, the last one stores the row.
    // The residual of a round's piece j: lane (q, c15) fetches its own share of pixel 4q + j - of z = cat(view i, partner) (64 channels =
    // 128 bytes each) the 16 bytes that hold channels 8 c15 .. 8 c15 + 7.  rq holds two rounds: 0, 1, later 2, 3.
    constexpr int SUBS = NCB / 2 >= 4 ? 2 : 1, PPS = (NCB / 2) / SUBS;
    constexpr int NSUB = 2 * 4 * SUBS;                       // sub-pieces of two rounds: 16 | 8
    constexpr int SLOT0 = 2 * NSTEP - NSUB;                  // they fill the LAST slots of a half stage (two slots per step): 8.. | 4..
    lane_row_t rq[2][4];
    lane_row_t ocur;
    auto res_load2 = [&](int rbase) __attribute__((always_inline)) {
        if (V11_ABL & 4) return;
        int le = lane;
        asm volatile("" : "+v"(le));
        const int c15e = le & 15, qe = le >> 4;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = rbase + rr;
                const int gy = y0 + 2 * w + (r >> 1), gyc = gy < H ? gy : H - 1;
                const int gx = x0 + 16 * (r & 1) + 4 * qe + j, gxc = gx < W ? gx : W - 1;
                const unsigned char* view = (RESM == 2 && c15e >= 8) ? resB : resA;
                const unsigned char* src = view + ((unsigned)((gyc * W + gxc) * 128) + (RESM == 2 ? (unsigned)((c15e & 7) * 16) : (unsigned)(c15e * LB)));
                rq[rr][j] = __builtin_nontemporal_load((const lane_row_t*)src);
            }
    };
    auto epi_sub = [&](auto r_c, auto j_c, auto ih_c) __attribute__((always_inline)) {
        constexpr int r = decltype(r_c)::value, j = decltype(j_c)::value, ih = decltype(ih_c)::value;
        const lane_row_t rv = rq[r & 1][j];
#pragma unroll
        for (int ii = 0; ii < PPS; ++ii) {
            const int i = ih * PPS + ii;
            float xa = acc[2 * i][r][j], xb = acc[2 * i + 1][r][j];
            xa = __builtin_amdgcn_fmed3f(xa, act_slope * xa, act_pick);
            xb = __builtin_amdgcn_fmed3f(xb, act_slope * xb, act_pick);
            if (RES) {
                const float ra = __uint_as_float(rv[i] << 16), rb = __uint_as_float(rv[i] & 0xffff0000u);
                if (RESM == 3) { xa = ra + res_alpha * xa; xb = rb + res_alpha * xb; }
                else { xa += ra; xb += rb; }
            }
            ocur[i] = pack2_bf16(xa, xb);
        }
        if constexpr (ih == SUBS - 1) {
            int le = lane;
            asm volatile("" : "+v"(le));
            const int c15e = le & 15, qe = le >> 4;
            const int gy = y0 + 2 * w + (r >> 1), gx = x0 + 16 * (r & 1) + 4 * qe + j;
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, (int)(hw * ROW), 0x00020000);
            // a pixel outside the image gets an offset the descriptor's range check drops: no branch around the store
            const unsigned voff = (unsigned)((gy * W + gx) * ROW + c15e * LB) | ((unsigned)(W - 1 - gx) & OOB) | (gy < H ? 0u : OOB);
            if (!(V11_ABL & 2)) {
                if constexpr (LB == 16) __builtin_amdgcn_raw_buffer_store_b128(ocur, rs_out, voff, 0, 2);      // nt: the next launch reads it from HBM anyway
                else __builtin_amdgcn_raw_buffer_store_b64(ocur, rs_out, voff, 0, 2);
            } else asm volatile("" :: "v"(ocur), "v"(voff));
        }
    };
    // sub-piece number u of the two rounds [rbase, rbase + 2)
    auto epi_slot = [&](auto u_c, auto rbase_c) __attribute__((always_inline)) {
        constexpr int u = decltype(u_c)::value, rbase = decltype(rbase_c)::value;
        if constexpr (u >= 0 && u < NSUB && !(V11_ABL & 4))
            epi_sub(std::integral_constant<int, rbase + u / (4 * SUBS)>{}, std::integral_constant<int, (u / SUBS) % 4>{}, std::integral_constant<int, u % SUBS>{});
    };

    // ---- DMA duties of a stage (c, tg), the same in every wave: the weights of the next stage, and under tap rows 0 / 1 pieces 0-2 / 3-4
    // of the next halo chunk.  Item `it` is issued from the gap behind the it-th step.
    struct StageDma { bool have_next, next_chunk; int c2, tg2, slot_w, cn, nbuf, tg; __amdgpu_buffer_rsrc_t rs_h; };
    auto stage_dma = [&](int c, int tg, int slot_r, bool more_tiles) __attribute__((always_inline)) -> StageDma {
        StageDma d;
        d.tg = tg;
        d.have_next = c < NCH - 1 || tg < 2 || more_tiles;
        d.next_chunk = c < NCH - 1 || more_tiles;
        d.tg2 = tg == 2 ? 0 : tg + 1;
        d.c2 = tg == 2 ? ((c + 1) & (NCH - 1)) : c;
        d.slot_w = slot_r ^ 1;
        d.cn = (c + 1) & (NCH - 1);
        d.nbuf = d.cn & 1;
        const size_t hb = (PAIR && d.cn >= 2) ? inB : inA;    // in the tile's last chunk these already are the next tile's views
        d.rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)(src0 + hb), 0, (int)img_bytes, 0x00020000);
        return d;
    };
    auto issue_item = [&](auto it_c, const StageDma& d) __attribute__((always_inline)) {
        constexpr int it = decltype(it_c)::value;
        if constexpr (it < NW3) { if (d.have_next) dma_w(d.c2, d.tg2, d.slot_w, it); }
        else if constexpr (it < N_ITEMS) {
            constexpr int k = it - NW3;
            if (d.next_chunk) {
                if (d.tg == 0) dma_halo(d.rs_h, d.cn, d.nbuf, k, hoff[k]);
                else if (d.tg == 1 && k < 2) dma_halo(d.rs_h, d.cn, d.nbuf, k + 3, hoff[k + 3]);
            }
        }
    };
    auto halo_out_of = [&](const StageDma& d) __attribute__((always_inline)) -> int {      // halo pieces the wave leaves in flight at the stage's end
        return d.next_chunk ? (d.tg == 0 ? 3 : d.tg == 1 ? n_in - 3 : 0) : 0;
    };

    bf16x8 fa[2][2], fb[2][4];
    auto rd = [&](bf16x8& dst, unsigned addr, int imm) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm));
    };

    // ---- an ordinary stage: NSTEP steps = 3 taps x NQ cout pairs, 8 MFMAs each ((k, pxb): cout block 2qt+k x pixel block pxb).
    // Hand-issued fragment reads with counted waits (conv3x3_v6.hip): prologue B0..B3(tap 0), A0(0), A1(0); step i: A0(i+1) after MFMA
    // 0, A1(i+1) after MFMA 1, and in the second step of a tap the next tap's B0..B3 after MFMAs 2..5; one DMA item behind a step.
    auto stage_full = [&](int tg, int slot_r, int buf, const StageDma& d) __attribute__((always_inline)) {
        const unsigned abase = a_off + (unsigned)(slot_r * WST);
        const unsigned boff = (unsigned)(buf * IN_BYTES + tg * ROWB);
        unsigned bcur[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) bcur[j][kx] = baddr0[j][kx] + boff;
        auto load_b1 = [&](int tap, int pxb) __attribute__((always_inline)) { rd(fb[tap & 1][pxb], bcur[pxb >> 1][tap], (pxb & 1) * 1024); };
        auto load_a1 = [&](int i, int k) __attribute__((always_inline)) { rd(fa[i & 1][k], abase, (i / NQ) * TAP_BYTES + (i % NQ) * 2048 + k * 1024); };
        if (BAL && w >= 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pxb = 0; pxb < 4; ++pxb) load_b1(0, pxb);
        load_a1(0, 0);
        load_a1(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NSTEP>([&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value;
            constexpr int qt = i % NQ, tap = i / NQ, bs = tap & 1;
            constexpr bool a_next = i + 1 < NSTEP;
            constexpr bool b_cur = qt == (NQ > 1 ? 1 : 0) && tap + 1 < 3;
            constexpr bool b_prev = i >= 1 && ((i - 1) % NQ) == (NQ > 1 ? 1 : 0) && (i - 1) / NQ + 1 < 3;
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
                acc[qt * 2 + k][pxb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[bs][pxb], fa[i & 1][k], acc[qt * 2 + k][pxb], 0, 0, 0);
                if (g < 2 && a_next) load_a1(i + 1, g);
                if (g >= 2 && g < 6 && b_cur) load_b1(tap + 1, g - 2);
                if (g == 7) issue_item(std::integral_constant<int, i>{}, d);
                if (BAL && g == 7 && i == (NSTEP * BAL) / 8 - 1 && w >= 4) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    // ---- half a stage: the same NSTEP steps for the two pixel blocks of ONE pixel row (HALF 0: row 2w, blocks 0, 1; HALF 1: row 2w + 1,
    // blocks 2, 3), 4 MFMAs each ((k, pp): cout block 2qt+k x block 2 HALF + pp).  Reads: prologue B0, B1(tap 0), A0(0), A1(0); step i:
    // A0(i+1) after MFMA 0, A1(i+1) after MFMA 1, in the second step of a tap the next tap's B0, B1 after MFMAs 2, 3.  PIECES: the
    // sub-pieces of epilogue rounds RBASE, RBASE + 1 behind MFMAs 1 and 3 of the last steps; ITEMS: the stage's DMA items, one per step.
    auto stage_half = [&](auto half_c, auto pieces_c, auto rbase_c, auto items_c, int tg, int slot_r, int buf, const StageDma& d) __attribute__((always_inline)) {
        constexpr int HALF = decltype(half_c)::value, RBASE = decltype(rbase_c)::value;
        constexpr bool PIECES = decltype(pieces_c)::value, ITEMS = decltype(items_c)::value;
        const unsigned abase = a_off + (unsigned)(slot_r * WST);
        const unsigned boff = (unsigned)(buf * IN_BYTES + tg * ROWB);
        unsigned bcur[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) bcur[kx] = baddr0[HALF][kx] + boff;
        auto load_b1 = [&](int tap, int pp) __attribute__((always_inline)) { rd(fb[tap & 1][pp], bcur[tap], pp * 1024); };
        auto load_a1 = [&](int i, int k) __attribute__((always_inline)) { rd(fa[i & 1][k], abase, (i / NQ) * TAP_BYTES + (i % NQ) * 2048 + k * 1024); };
        load_b1(0, 0);
        load_b1(0, 1);
        load_a1(0, 0);
        load_a1(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NSTEP>([&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value;
            constexpr int qt = i % NQ, tap = i / NQ, bs = tap & 1;
            constexpr bool a_next = i + 1 < NSTEP;
            constexpr bool b_cur = qt == (NQ > 1 ? 1 : 0) && tap + 1 < 3;
            constexpr bool b_prev = i >= 1 && ((i - 1) % NQ) == (NQ > 1 ? 1 : 0) && (i - 1) / NQ + 1 < 3;
            // reads allowed to be outstanding before MFMA 0 (needs A0(i); younger: A1(i), the B pair of the previous step - needed at once
            // when NQ == 2) and before MFMA 2 (needs A1(i); younger: that B pair, A0(i+1), A1(i+1))
            constexpr int n0 = (b_prev && NQ == 2) ? 0 : 1 + (b_prev ? 2 : 0);
            constexpr int n2 = (b_prev ? 2 : 0) + (a_next ? 2 : 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k = g >> 1, pp = g & 1;
                if (g == 0) {
                    if (n0 == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[i & 1][0]), "+v"(fb[bs][0]), "+v"(fb[bs][1]));
                    else if (n0 == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa[i & 1][0]), "+v"(fb[bs][0]), "+v"(fb[bs][1]));
                    else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[i & 1][0]), "+v"(fb[bs][0]), "+v"(fb[bs][1]));
                } else if (g == 2) {
                    if (n2 == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[i & 1][1]));
                    else if (n2 == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[i & 1][1]));
                    else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[i & 1][1]));
                }
                acc[qt * 2 + k][2 * HALF + pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[bs][pp], fa[i & 1][k], acc[qt * 2 + k][2 * HALF + pp], 0, 0, 0);
                if (g < 2 && a_next) load_a1(i + 1, g);
                if (g >= 2 && b_cur) load_b1(tap + 1, g - 2);
                if constexpr (PIECES) {
                    if (g == 1) epi_slot(std::integral_constant<int, 2 * i - SLOT0>{}, rbase_c);
                    if (g == 3) epi_slot(std::integral_constant<int, 2 * i + 1 - SLOT0>{}, rbase_c);
                }
                if constexpr (ITEMS) { if (g == 3) issue_item(std::integral_constant<int, i>{}, d); }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    // ---- prologue: weights of stage 0 -> slot 0, halo chunk 0 of the first tile -> buffer 0
    if (tid < COUT) bias_lds[tid] = p.bias[tid];
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src0 + inA), 0, (int)img_bytes, 0x00020000);
#pragma unroll
        for (int t3 = 0; t3 < NW3; ++t3) dma_w(0, 0, 0, t3);
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) dma_halo(rs, 0, 0, jj, hoff[jj]);
    }
    wait_vm<0>();
    wg_barrier();

    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 1> I1{};
    constexpr std::integral_constant<int, 2> I2{};
    constexpr std::integral_constant<int, 4> I4{};
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};

    for (int tl = 0; tl < ntl; ++tl) {
        const bool more_tiles = tl + 1 < ntl;
        unsigned nxt_m = cur_m, nxt_t = cur_t;
        next_tile(nxt_m, nxt_t);
        size_t nxA = inA, nxB = inB;
        if (more_tiles) in_bases(nxt_m, nxA, nxB);

        // ---- stage 0 (chunk 0, tap row 0, weight slot 0, halo buffer 0)
        int st0 = 1;
        if (tl == 0) {
            geometry(cur_m, cur_t);
            init_acc(I0, I4);
            st0 = 0;                                        // the first tile has nothing to finish: an ordinary stage
        } else {
            const StageDma d = stage_dma(0, 0, 0, more_tiles);
            // row 2w of the new tile, with rounds 2, 3 of the previous tile (whose geometry is still in place) behind its MFMAs
            init_acc(I0, I2);
            stage_half(I0, YES, I2, NO, 0, 0, 0, d);
            geometry(cur_m, cur_t);
            init_acc(I2, I4);
            // row 2w + 1, with the stage's DMA items: none is younger than a residual load that is still to be used
            stage_half(I1, NO, I0, YES, 0, 0, 0, d);
            wait_vm_rt(halo_out_of(d));
            wg_barrier();
        }
        // ---- stages st0 .. 10
        int c = 0, tg = st0;
        for (int st = st0; st < 3 * NCH - 1; ++st) {
            if (c == NCH - 1 && tg == 0) {                  // this tile's last halo chunk is on its way: from here on the DMA state describes the next tile
                inA = nxA; inB = nxB;
                if (more_tiles) tile_offsets(nxt_t);
            }
            const int slot_r = (c + tg) & 1;
            const StageDma d = stage_dma(c, tg, slot_r, more_tiles);
            stage_full(tg, slot_r, c & 1, d);
            // the next stage's weights (and every older access) have landed once only this stage's halo pieces are outstanding
            wait_vm_rt(halo_out_of(d));
            wg_barrier();
            if (++tg == 3) { tg = 0; ++c; }
        }
        // ---- stage 11 (chunk 3, tap row 2, slot 1, buffer 1): the weights of the next tile's stage 0 and the residual of rounds 0, 1 are
        // requested first; row 2w; then row 2w + 1 with rounds 0, 1 behind its MFMAs; the residual of rounds 2, 3 last
        {
            const StageDma d = stage_dma(NCH - 1, 2, 1, more_tiles);
            static_for<NW3>([&](auto it_c) __attribute__((always_inline)) { issue_item(it_c, d); });
            if (RES) res_load2(0);
            stage_half(I0, NO, I0, NO, 2, 1, 1, d);
            // the weights are older than the eight residual loads: landed once only those are outstanding.  (Waited for here, not at the
            // stage's end: by then the youngest access is a store of this half's epilogue.)
            if (RES) wait_vm<8>(); else wait_vm<0>();
            stage_half(I1, YES, I0, NO, 2, 1, 1, d);
            if (RES) res_load2(2);
            if (V11_ABL & 4) {                              // (keep the MFMAs alive)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int pxb = 0; pxb < 4; ++pxb) asm volatile("" :: "v"(acc[cb][pxb]));
            }
            wg_barrier();
        }
        cur_m = nxt_m; cur_t = nxt_t;
    }
    // ---- rounds 2, 3 of the last tile
    static_for<NSUB>([&](auto u_c) __attribute__((always_inline)) { epi_slot(u_c, I2); });
    wait_vm<0>();                                           // nothing of this workgroup may still be in flight when it ends
}

template <int COUT, int RESM, bool PAIR>
int launch_v11(const ConvParams& p, long grid, hipStream_t stream) {
    typedef G11<COUT> GEO;
    static_assert(GEO::LDS_BYTES <= 160 * 1024, "LDS budget");
    { const int rc_lds = hrn_allow_lds((const void*)conv3x3_v11_kernel<COUT, RESM, PAIR>, GEO::LDS_BYTES); if (rc_lds) return rc_lds; }
    hipLaunchKernelGGL((conv3x3_v11_kernel<COUT, RESM, PAIR>), dim3((unsigned)grid), dim3(512), GEO::LDS_BYTES, stream, p);
    HRN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int hrn_launch_conv3x3_v11(int cout, const ConvParams& p, hipStream_t stream) {
    if (p.scale || p.relu) return -100;
    if (cout != 64 && cout != 128) return -100;
    if (cout == 128 && p.res_mode != 0 && p.res_mode != 2) return -100;
    if (cout == 64 && ((p.res_mode != 0 && p.res_mode != 3) || p.in_pair)) return -100;
    if ((p.in_pair || p.res_mode == 2) && p.pair_h <= 0) return -100;
    if (p.res_mode == 3 && (p.out_h <= 0 || !p.res)) return -100;
    const long tiles = (long)((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH);
    const long total = tiles * p.M;
    HRN_CHECK(total > 0, -2, "conv3x3_v11: bad tile count %ld", total);
    if (total >= (1L << 30) || (long)p.H * p.W * 256 >= (1L << 31)) return -100;     // 32-bit tile / in-image byte arithmetic
    long grid = hrn_device_cus();
    if (total < grid) grid = total;
    if (grid >= 8) grid &= ~7L;
    const double px = (double)p.M * p.H * p.W;
    const char* fam = cout == 128 ? (p.res_mode ? "conv3x3_bf16_128x128+res" : "conv3x3_bf16_128x128")
                                  : (p.res_mode ? "conv3x3_bf16_128x64+res" : "conv3x3_bf16_128x64");
    HrnProfScope prof(fam, 2.0 * 128 * cout * 9 * px, px * 2 * (128 + cout + (p.res_mode ? cout : 0)), stream);
    if (cout == 128) {
        if (p.in_pair) return p.res_mode ? launch_v11<128, 2, true>(p, grid, stream) : launch_v11<128, 0, true>(p, grid, stream);
        return p.res_mode ? launch_v11<128, 2, false>(p, grid, stream) : launch_v11<128, 0, false>(p, grid, stream);
    }
    return p.res_mode ? launch_v11<64, 3, false>(p, grid, stream) : launch_v11<64, 0, false>(p, grid, stream);
}
