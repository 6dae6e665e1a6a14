// conv3x3 128 -> {128, 64}, bf16: the three layers of a fusion level (HRNet.py:90-97, :113-131) - 61 % of the forward FLOPs.
// The skeleton of conv3x3_v6.hip (512-pixel tiles, 8 MFMA waves, two per SIMD, descriptor-based LDS-DMA issued from the MFMA
// gaps, 2-slot weight ring, LDS-staged coalesced epilogue) on v_mfma_f32_32x32x16_bf16.
// Why the other MFMA shape: the stamps of conv3x3_v6 (profiles/r02_final_v6_stamps.txt) show the kernel bound by the SIMD's
// vector-issue port, not by the matrix pipe: an MFMA of either shape holds the port for 8 cycles, so 16x16x32 spends 8 of
// every 16 pipe cycles on MFMA issue alone and the fragment reads, DMA issues and waits of BOTH waves of the SIMD queue up
// behind them (a wave's 96 MFMAs of a stage took 2.7 k cycles even with priority; 1.5 k would be the pipe's rate).  With
// 32x32x16 the same FLOPs need half the MFMA instructions and exactly the same fragment reads (0.75 per MFMA of 32 pipe
// cycles), which leaves the port half idle.
//   per wave   pixel rows 2w, 2w+1 (two blocks of 32 pixels) x COUT output channels (COUT / 32 blocks): COUT accumulator
//              registers; a k-step (16 input channels of one tap) = NCB A + 2 B fragment reads for 2 NCB MFMAs, reads of step
//              i+1 issued one per MFMA gap of step i (hand-written ds_read_b128 with counted lgkmcnt, see conv3x3_v6.hip)
//   LDS images rows of 64 bytes (32 input channels) per cout / halo pixel, 16-byte chunk c stored at c ^ ((row >> 2) & 3)
//              (conflict-free ds_read_b128); applied on the DMA source side and on the read
//   B address  one register per (halo row 2w + j, tap column kx); k-step 1 is the same address with bit 5 flipped (one v_xor)
//   epilogue   four rounds per wave, round = (pixel row pb, channel half h): 32 pixels x 64 channels = 4 KB of staging rows
//              (128 B per pixel, segments XOR-ed with (pixel >> 1) & 7).  Residual: round 0 by LDS-DMA during the tile's last
//              stage, later rounds by lane-contiguous loads one round ahead (an instruction covers 8 whole 128-byte lines),
//              read back in the accumulator layout (8 bytes = one accumulator quad); results return the same way and leave
//              as whole lines.
// RESM: 0 none | 2 the pair gather z (COUT = 128: t2 = z + PReLU(conv(t1))) | 3 s_i + alpha_partner * f into the view stack
// (COUT = 64, HRNet.py:123-131).  PAIR: the input is the pair gather cat(view i, partner) of the view stack.
// LDS (COUT = 128): 2 x 24,576 (weights) + 2 x 39,936 (halo) + 512 (bias) + 8 x 4,096 (staging) = 162,304 B.
// Ordering rules (guide, "Pipelining across barriers"): a wave waits for its own DMAs with a counted vmcnt BEFORE the
// barrier that precedes the stage reading them; a buffer is re-filled only after a barrier every reader of its previous
// contents has passed.  vmcnt counts in issue order, so a stage issues its weights first and its halo pieces last: waiting
// until only the halo pieces are outstanding retires the weights and everything older.
#include <type_traits>
#include "conv3x3.h"

// Timing-only ablations (tools/v6_abl.sh v7 BITS; results are WRONG when set): 1 no MFMA | 2 no epilogue | 4 no DMA | 32 no fragment reads
#ifndef V7_ABL
#define V7_ABL 0
#endif

namespace {

constexpr int T7_H = 16, T7_W = 32;
constexpr int HW7 = T7_W + 2;                              // halo width 34
constexpr int NPIX7 = (T7_H + 2) * HW7;                    // 612 halo pixels
constexpr int N_IN7 = (NPIX7 * 64 + 1023) / 1024;          // 39 DMA pieces of 1 KB per 32-channel halo chunk
constexpr int IN_BYTES7 = N_IN7 * 1024;                    // 39,936
constexpr unsigned OOB7 = 0x80000000u;                     // byte offset no descriptor of this kernel covers

template <int COUT> struct G7 {
    static constexpr int NCB = COUT / 32;                  // cout blocks of 32 per wave
    static constexpr int TAP_BYTES = COUT * 64;            // one tap x 32 cin
    static constexpr int WST = 3 * TAP_BYTES;              // one stage: 24,576 | 12,288
    static constexpr int W_PIECES = WST / 1024;            // 24 | 12
    static constexpr int OFF_IN = 2 * WST;
    static constexpr int OFF_BIAS = OFF_IN + 2 * IN_BYTES7;
    static constexpr int OFF_STG = OFF_BIAS + 512;
    static constexpr int STG_WAVE = 4096;                  // one round: 32 pixels x 64 channels
    static constexpr int NRND = COUT / 32;                 // rounds per wave: (2 pixel rows) x (COUT / 64 channel halves) = 4 | 2
    static constexpr int LDS_BYTES = OFF_STG + 8 * STG_WAVE;
};

typedef __attribute__((address_space(3))) void* lds_ptr7;

__device__ __forceinline__ float raw_max7(float a, float b) {
    float y;
    asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
    return y;
}
template <int N> __device__ __forceinline__ void wait_vm7() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm7_rt(int n) {       // n is wave-uniform, 0..5
    switch (n) {
        case 1: wait_vm7<1>(); break;
        case 2: wait_vm7<2>(); break;
        case 3: wait_vm7<3>(); break;
        case 4: wait_vm7<4>(); break;
        case 5: wait_vm7<5>(); break;
        default: wait_vm7<0>(); break;
    }
}
__device__ __forceinline__ void barrier7() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N, int I = 0, class F> __device__ __forceinline__ void static_for7(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for7<N, I + 1>(f); }
}

template <int COUT, int RESM, bool PAIR>
__global__ __launch_bounds__(512, 2) void conv3x3_v7_kernel(const ConvParams p) {
    typedef G7<COUT> GEO;
    constexpr int NCB = GEO::NCB, WST = GEO::WST, TAP_BYTES = GEO::TAP_BYTES, OFF_IN = GEO::OFF_IN, NRND = GEO::NRND;
    constexpr int OPIX = COUT * 2;                          // bytes per output pixel
    constexpr bool RES = RESM != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bias_lds = (float*)(smem + GEO::OFF_BIAS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int H = p.H, W = p.W;
    const unsigned hw = (unsigned)(H * W);
    const unsigned tiles_x = (W + T7_W - 1) / T7_W;
    const unsigned tiles_y = (H + T7_H - 1) / T7_H;
    const unsigned tiles = tiles_x * tiles_y;
    const unsigned total = tiles * (unsigned)p.M;
    const unsigned G = gridDim.x;
    const unsigned bid = blockIdx.x;
    const unsigned slot0 = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;      // each XCD walks a contiguous run of tiles
    if (slot0 >= total) return;
    const int ntl = (int)((total - slot0 + G - 1) / G);
    unsigned cur_m = slot0 / tiles, cur_t = slot0 - cur_m * tiles;
    const unsigned step_m = G / tiles, step_t = G - step_m * tiles;
    constexpr bool in_pair = PAIR;
    constexpr unsigned in_pitch = in_pair ? 128u : 256u;
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
            a = b = (size_t)m * hw * 256;
        }
    };
    // ---- per-lane byte offsets of this wave's halo pieces for tile t (pieces j = w + 8 jj < 39; lane i -> halo pixel
    // j*16 + (i >> 2), physical 16-byte chunk i & 3 = logical chunk ^ ((pixel >> 2) & 3)); invalid pixels -> OOB7, which the
    // descriptor's range check turns into zeros in LDS
    unsigned hoff[5];
    auto tile_offsets = [&](unsigned t) __attribute__((always_inline)) {
        const int ty = t / tiles_x;
        const int y0 = ty * T7_H, x0 = (t - ty * tiles_x) * T7_W;
        int lq = lane;
        asm volatile("" : "+v"(lq));                        // keep the per-piece geometry out of long-lived registers
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int pix = (w + 8 * jj) * 16 + (lq >> 2);
            const int lc = (lq & 3) ^ ((pix >> 2) & 3);
            const int py = pix / HW7, px = pix - py * HW7;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool ok = pix < NPIX7 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            hoff[jj] = ok ? (unsigned)(gy * W + gx) * in_pitch + (unsigned)(lc * 16) : OOB7;
        }
    };
    // one halo piece: chunk c (32 channels = 64 bytes of a pixel) of the image behind `rs` -> input buffer `buf`
    auto dma_halo = [&](__amdgpu_buffer_rsrc_t rs, int c, int buf, int jj) __attribute__((always_inline)) {
        const int j = w + 8 * jj;
        if (j < N_IN7) {
            const unsigned soff = in_pair ? (unsigned)((c & 1) * 64) : (unsigned)(c * 64);
            if (!(V7_ABL & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr7)(smem + OFF_IN + buf * IN_BYTES7 + j * 1024), 16, hoff[jj], soff, 0, 0);
        }
    };
    // one weight piece of stage (c, tg): piece qq = (tap kx = qq / GPT, cout group jb = qq % GPT): 16 couts x 64 bytes;
    // lane i -> cout 16 jb + (i >> 2), physical chunk i & 3 = logical (i & 3) ^ ((cout >> 2) & 3)
    constexpr int GPT = COUT / 16;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, 9 * 128 * COUT * 2, 0x00020000);
    const unsigned w_lane_off = (unsigned)((lane >> 2) * 128 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    auto dma_w = [&](int c, int tg, int slot_, int t3) __attribute__((always_inline)) {
        const int qq = w + 8 * t3;
        if (qq < GEO::W_PIECES) {
            const int kx = qq / GPT, jb = qq - kx * GPT;
            const unsigned soff = (unsigned)(((c >> 1) * 9 + tg * 3 + kx) * (COUT * 128) + (c & 1) * 64 + jb * 2048);
            if (!(V7_ABL & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr7)(smem + slot_ * WST + kx * TAP_BYTES + jb * 1024), 16, w_lane_off, soff, 0, 0);
        }
    };
    constexpr int NW3 = (GEO::W_PIECES + 7) / 8;            // weight pieces a wave issues per stage: up to 3 | 2
    const int n_in = w < (N_IN7 & 7) ? (N_IN7 >> 3) + 1 : (N_IN7 >> 3);      // halo pieces of this wave per chunk: 5 (wave 7: 4)

    const bool has_slope = p.slope != nullptr;
    const float slope = has_slope ? p.slope[0] : 0.f;
    const bool slope01 = slope >= 0.f && slope <= 1.f;

    // fragment addresses.  A: cout row cb*32 + r, k-step ks -> a_off[ks] + slot*WST + kx*TAP + cb*2048.  B: one register per (halo
    // row 2w + j, tap column kx) for k-step 0 (logical chunk hh); k-step 1 (chunk 2 + hh) is the same address with bit 5 flipped.
    // The input buffer's offset is folded in once per chunk.
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned a_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[ks] = lds0 + (unsigned)(r * 64 + (((ks * 2 + hh) ^ ((r >> 2) & 3)) << 4));
    unsigned baddr[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int pix = (2 * w + j) * HW7 + r + kx;
            baddr[j][kx] = lds0 + (unsigned)OFF_IN + (unsigned)(pix * 64 + ((hh ^ ((pix >> 2) & 3)) << 4));
        }

    f32x16 acc[NCB][2];                                     // [cout block of 32][pixel row]

    // ---- prologue: weights of stage 0, halo chunk 0 of the first tile
    if (tid < COUT) bias_lds[tid] = p.bias[tid];
    size_t inA, inB;
    in_bases(cur_m, inA, inB);
    tile_offsets(cur_t);
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src0 + inA), 0, (int)img_bytes, 0x00020000);
#pragma unroll
        for (int t3 = 0; t3 < NW3; ++t3) dma_w(0, 0, 0, t3);
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) dma_halo(rs, 0, 0, jj);
    }
    wait_vm7<0>();
    barrier7();

    for (int tl = 0; tl < ntl; ++tl) {
        const bool more_tiles = tl + 1 < ntl;
        unsigned nxt_t = cur_t + step_t, nxt_m = cur_m + step_m;
        if (nxt_t >= tiles) { nxt_t -= tiles; ++nxt_m; }
        size_t nxA = inA, nxB = inB;
        if (more_tiles) in_bases(nxt_m, nxA, nxB);
        // the accumulators start at the bias (element 4g + j of block cb = channel cb*32 + 8g + 4hh + j)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *(const f32x4*)(bias_lds + cb * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[cb][0][4 * g + j] = b[j]; acc[cb][1][4 * g + j] = b[j]; }
            }
        // geometry of this tile's outputs (used by the residual prefetch and the epilogue)
        const int ty_ = cur_t / tiles_x;
        const int y0 = ty_ * T7_H, x0 = (cur_t - ty_ * tiles_x) * T7_W;
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
            outp = (unsigned char*)p.out + oimg * hw * OPIX;
        }
        // round rd = (pixel row pb = rd / (NRND / 2), channel half h = rd % (NRND / 2)); its residual is 32 pixels x 128 bytes of
        // ONE view (RESM 2: half 0 = view i, half 1 = its partner; RESM 3: s_i): piece k, lane i <-> pixel pp = 8k + (i >> 3) of the
        // row, 16-byte segment i & 7.  `swizzled`: the lane fetches the segment that belongs at its position of the swizzled
        // staging row (LDS-DMA writes lane i's bytes to position i).
        auto res_src = [&](int rd, int k, bool swizzled) __attribute__((always_inline)) -> const unsigned char* {
            const int pb = rd / (NRND / 2), h = rd % (NRND / 2);
            const int gy = y0 + 2 * w + pb, gyc = gy < H ? gy : H - 1;
            int lq = lane;
            asm volatile("" : "+v"(lq));
            const int pp = 8 * k + (lq >> 3);
            const int s = swizzled ? (lq & 7) ^ ((pp >> 1) & 7) : (lq & 7);
            const int gx = x0 + pp, gxc = gx < W ? gx : W - 1;
            const unsigned char* view = (RESM == 2 && h == 1) ? resB : resA;
            return view + (unsigned)((gyc * W + gxc) * 128 + s * 16);
        };
        auto res_dma0 = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)res_src(0, k, true),
                                                 (lds_ptr7)(smem + GEO::OFF_STG + w * GEO::STG_WAVE + k * 1024), 16, 0, 0);
        };

        for (int c = 0; c < 4; ++c) {
            if (c > 0 || tl > 0) {      // chunk c sits in input buffer c & 1: move the B addresses over from the other buffer
                const unsigned d = (c & 1) ? (unsigned)IN_BYTES7 : (unsigned)-IN_BYTES7;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) baddr[j][kx] += d;
            }
            if (c == 3) {        // this tile's last halo chunk is on its way: from here on the DMA state describes the next tile
                inA = nxA; inB = nxB;
                if (more_tiles) tile_offsets(nxt_t);
            }
            auto stage = [&](auto tg_c) __attribute__((always_inline)) {
                constexpr int tg = decltype(tg_c)::value;
                const int slot_r = (c + tg) & 1;                                        // ring slot this stage reads
                const bool have_next = c < 3 || tg < 2 || more_tiles;                   // there is a stage s+1
                const bool next_chunk = c < 3 || more_tiles;                            // there is a halo chunk after this one
                const int tg2 = (tg + 1) % 3, c2 = (c + (tg + 1) / 3) & 3;
                const int cn = (c + 1) & 3;
                const size_t hb = (in_pair && cn >= 2) ? inB : inA;      // at c == 3 these already are the next tile's views
                const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)(src0 + hb), 0, (int)img_bytes, 0x00020000);
                // DMA item `it` of this stage, issued from the gap behind k-step it: weights of stage s+1 first, then (tg 0: pieces
                // jj 0-2, tg 1: pieces 3-4) of the next halo chunk
                auto issue_item = [&](int it) __attribute__((always_inline)) {
                    if (it < NW3) { if (have_next) dma_w(c2, tg2, slot_r ^ 1, it); }
                    else if (tg == 0 && it < NW3 + 3) { if (next_chunk) dma_halo(rs_h, cn, cn & 1, it - NW3); }
                    else if (tg == 1 && it < NW3 + 2) { if (next_chunk) dma_halo(rs_h, cn, cn & 1, it - NW3 + 3); }
                };
                constexpr int N_ITEMS = NW3 + (tg == 0 ? 3 : tg == 1 ? 2 : 0);
                static_assert(N_ITEMS <= 6, "one DMA item per k-step");
                int halo_out = 0;                                                      // halo pieces this wave leaves in flight
                if (next_chunk) halo_out = tg == 0 ? 3 : tg == 1 ? n_in - 3 : 0;
                if (RES && tg == 2 && c == 3 && !(V7_ABL & 2)) res_dma0();
                // ---- 3 taps x 2 k-steps; fragment reads one step ahead of their MFMAs, one read per MFMA gap, in the order
                // A[0..NCB-1], B0, B1; the MFMAs run (cb, pb) = (0,0) (0,1) (1,0) ...  LDS reads return in order: before (0,0) of
                // step i, A[*](i) and B0(i) are back once at most B1(i) is outstanding: lgkmcnt(1); before (0,1), B1(i): only A0(i+1),
                // issued in gap 0, is younger: lgkmcnt(1) (0 in the last step, which issues nothing).
                const unsigned abase0 = a_off[0] + (unsigned)(slot_r * WST), abase1 = a_off[1] + (unsigned)(slot_r * WST);
                bf16x8 fa[2][NCB], fb[2][2];
                auto rd = [&](bf16x8& dst, unsigned addr, int imm) __attribute__((always_inline)) {
                    if (V7_ABL & 32) asm volatile("; no read" : "=v"(dst) : "v"(addr), "n"(imm));
                    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm));
                };
                auto load_part = [&](int i, int part) __attribute__((always_inline)) {
                    const int s_ = i & 1, kx = i >> 1, ks = i & 1;
                    if (part < NCB) {
                        rd(fa[s_][part], ks ? abase1 : abase0, kx * TAP_BYTES + part * 2048);
                    } else {
                        unsigned a = baddr[part - NCB + tg][kx];
                        if (ks) a ^= 32u;
                        rd(fb[s_][part - NCB], a, 0);
                    }
                };
#pragma unroll
                for (int part = 0; part < NCB + 2; ++part) load_part(0, part);
                __builtin_amdgcn_sched_barrier(0);
                static_for7<6>([&](auto i_c) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_c)::value;
                    constexpr int s_ = i & 1;
                    constexpr bool more = i + 1 < 6;
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int pb = 0; pb < 2; ++pb) {
                            const int g = cb * 2 + pb;              // MFMA gap index
                            if (g == 0) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa[s_][0]), "+v"(fb[s_][0]));
                            else if (g == 1) {
                                if (more) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fb[s_][1]));
                                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[s_][1]));
                            } else if (pb == 0) asm volatile("" : "+v"(fa[s_][cb]));
                            if (V7_ABL & 1) asm volatile("" : "+v"(acc[cb][pb][0]) : "v"(fa[s_][cb]), "v"(fb[s_][pb]));   // (no 512-bit asm operands: the host pass rejects them)
                            else acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][cb], fb[s_][pb], acc[cb][pb], 0, 0, 0);
                            if (more && g < NCB + 2) load_part(i + 1, g);
                            if (g == 2 * NCB - 1 && i < N_ITEMS) issue_item(i);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                });
                // stage s+1's weights (and every older DMA) have landed once only this stage's halo pieces are outstanding
                wait_vm7_rt(halo_out);                                   // tg 2: 0 (the residual DMA is older than the weights)
                if (tg == 2 && c == 3 && (V7_ABL & 2)) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int e = 0; e < 16; ++e) { asm volatile("" :: "v"(acc[cb][0][e])); asm volatile("" :: "v"(acc[cb][1][e])); }
                }
                if (tg == 2 && c == 3 && !(V7_ABL & 2)) {
                    // ---- epilogue of this tile: registers, global memory and this wave's staging rows only
                    unsigned char* stg = smem + GEO::OFF_STG + w * GEO::STG_WAVE;
                    // round 1 now, round rd + 2 when round rd has retired its accumulators: more at once do not fit beside the
                    // accumulators (hipcc then spills freshly loaded pieces, i.e. waits for them on the spot)
                    u32x4 rq[NRND][4];
                    auto res_load = [&](int rd_) __attribute__((always_inline)) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) rq[rd_][k] = *(const u32x4*)res_src(rd_, k, false);
                    };
                    if (RES && NRND > 1) res_load(1);
                    int le = lane;
                    asm volatile("" : "+v"(le));                    // every lane-derived address below is formed here, per tile
                    const int re = le & 31, he = le >> 5;
                    const int key = (re >> 1) & 7;                  // a pixel's staging row: 8 segments of 16 bytes, segment s at s ^ key
                    auto epilogue = [&](auto act_c) __attribute__((always_inline)) {
                        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
                        for (int rd_ = 0; rd_ < NRND; ++rd_) {
                            const int pb = rd_ / (NRND / 2), h = rd_ % (NRND / 2);
                            const int gy = y0 + 2 * w + pb;
                            if (RES && rd_ > 0) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const int pp = 8 * k + (le >> 3), s = le & 7;
                                    *(u32x4*)(stg + pp * 128 + ((s ^ ((pp >> 1) & 7)) << 4)) = rq[rd_][k];
                                }
                            }
                            // batched: all residual cells first, then the arithmetic, then all result cells, then the row pieces (with an
                            // LDS-DMA in the kernel hipcc answers every LDS read's first use with lgkmcnt(0): one round trip per batch).
                            // cell (cbl, g): the accumulator quad 4g..4g+3 of cout block 2h + cbl = channels 32 cbl + 8g + 4hh .. + 3 of the half
                            unsigned cell[2][4];
#pragma unroll
                            for (int cbl = 0; cbl < 2; ++cbl)
#pragma unroll
                                for (int g = 0; g < 4; ++g) cell[cbl][g] = (unsigned)(re * 128 + (((4 * cbl + g) ^ key) << 4) + he * 8);
                            u32x2 rr[2][4];
                            if (RES) {
#pragma unroll
                                for (int cbl = 0; cbl < 2; ++cbl)
#pragma unroll
                                    for (int g = 0; g < 4; ++g) rr[cbl][g] = *(const u32x2*)(stg + cell[cbl][g]);
                            }
                            u32x2 o[2][4];
#pragma unroll
                            for (int cbl = 0; cbl < 2; ++cbl)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    float x[4];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) x[e] = acc[2 * h + cbl][pb][4 * g + e];
                                    if (ACT == 1) {
#pragma unroll
                                        for (int e = 0; e < 4; ++e) x[e] = raw_max7(x[e], slope * x[e]);
                                    } else if (ACT == 2) {
#pragma unroll
                                        for (int e = 0; e < 4; ++e) x[e] = x[e] >= 0.f ? x[e] : slope * x[e];
                                    }
                                    if (RES) {
                                        const u32x2 q2 = rr[cbl][g];
                                        const float r0 = __uint_as_float(q2[0] << 16), r1 = __uint_as_float(q2[0] & 0xffff0000u);
                                        const float r2 = __uint_as_float(q2[1] << 16), r3 = __uint_as_float(q2[1] & 0xffff0000u);
                                        if (RESM == 3) { x[0] = r0 + res_alpha * x[0]; x[1] = r1 + res_alpha * x[1]; x[2] = r2 + res_alpha * x[2]; x[3] = r3 + res_alpha * x[3]; }
                                        else { x[0] += r0; x[1] += r1; x[2] += r2; x[3] += r3; }
                                    }
                                    o[cbl][g][0] = pack2_bf16(x[0], x[1]);
                                    o[cbl][g][1] = pack2_bf16(x[2], x[3]);
                                }
#pragma unroll
                            for (int cbl = 0; cbl < 2; ++cbl)
#pragma unroll
                                for (int g = 0; g < 4; ++g) *(u32x2*)(stg + cell[cbl][g]) = o[cbl][g];
                            u32x4 vv[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int pp = 8 * k + (le >> 3), s = le & 7;
                                vv[k] = *(const u32x4*)(stg + pp * 128 + ((s ^ ((pp >> 1) & 7)) << 4));
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int pp = 8 * k + (le >> 3), s = le & 7;
                                const int gx = x0 + pp;
                                if (gy < H && gx < W) *(u32x4*)(outp + (unsigned)((gy * W + gx) * OPIX + h * 128 + s * 16)) = vv[k];
                            }
                            if (RES && rd_ + 2 < NRND) res_load(rd_ + 2);     // one round of work between a fetch and its use
                        }
                    };
                    if (!has_slope) epilogue(std::integral_constant<int, 0>{});
                    else if (slope01) epilogue(std::integral_constant<int, 1>{});
                    else epilogue(std::integral_constant<int, 2>{});
                }
                barrier7();
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            stage(std::integral_constant<int, 2>{});
        }
        cur_m = nxt_m; cur_t = nxt_t;
    }
    wait_vm7<0>();                                          // nothing of this workgroup may still be in flight when it ends
}

template <int COUT, int RESM, bool PAIR>
int launch_v7(const ConvParams& p, long grid, hipStream_t stream) {
    typedef G7<COUT> GEO;
    static_assert(GEO::LDS_BYTES <= 160 * 1024, "LDS budget");
    { const int rc_lds = hrn_allow_lds((const void*)conv3x3_v7_kernel<COUT, RESM, PAIR>, GEO::LDS_BYTES); if (rc_lds) return rc_lds; }
    hipLaunchKernelGGL((conv3x3_v7_kernel<COUT, RESM, PAIR>), dim3((unsigned)grid), dim3(512), GEO::LDS_BYTES, stream, p);
    HRN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// bf16, 128 input channels.  COUT = 128: residual none or the pair gather (res_mode 2); COUT = 64: none or the alpha residual into
// the view stack (res_mode 3).  Returns -100 when not applicable.
int hrn_launch_conv3x3_v7(int cout, const ConvParams& p, hipStream_t stream) {
    if (p.scale || p.relu) return -100;
    if (cout != 64 && cout != 128) return -100;
    if (cout == 128 && p.res_mode != 0 && p.res_mode != 2) return -100;
    if (cout == 64 && ((p.res_mode != 0 && p.res_mode != 3) || p.in_pair)) return -100;
    if ((p.in_pair || p.res_mode == 2) && p.pair_h <= 0) return -100;
    if (p.res_mode == 3 && (p.out_h <= 0 || !p.res)) return -100;
    const long tiles = (long)((p.W + T7_W - 1) / T7_W) * ((p.H + T7_H - 1) / T7_H);
    const long total = tiles * p.M;
    HRN_CHECK(total > 0, -2, "conv3x3_v7: bad tile count %ld", total);
    if (total >= (1L << 30) || (long)p.H * p.W * 256 >= (1L << 31)) return -100;     // 32-bit tile / in-image byte arithmetic
    long grid = hrn_device_cus();
    if (total < grid) grid = total;
    if (grid >= 8) grid &= ~7L;
    const double px = (double)p.M * p.H * p.W;
    const char* fam = cout == 128 ? (p.res_mode ? "conv3x3_bf16_128x128+res" : "conv3x3_bf16_128x128")
                                  : (p.res_mode ? "conv3x3_bf16_128x64+res" : "conv3x3_bf16_128x64");
    HrnProfScope prof(fam, 2.0 * 128 * cout * 9 * px, px * 2 * (128 + cout + (p.res_mode ? cout : 0)), stream);
    if (cout == 128) {
        if (p.in_pair) return p.res_mode ? launch_v7<128, 2, true>(p, grid, stream) : launch_v7<128, 0, true>(p, grid, stream);
        return p.res_mode ? launch_v7<128, 2, false>(p, grid, stream) : launch_v7<128, 0, false>(p, grid, stream);
    }
    return p.res_mode ? launch_v7<64, 3, false>(p, grid, stream) : launch_v7<64, 0, false>(p, grid, stream);
}
