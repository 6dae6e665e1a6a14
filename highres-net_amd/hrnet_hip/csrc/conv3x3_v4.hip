// conv3x3 128 -> {128, 64}, bf16: the three layers of a fusion level (HRNet.py:90-97, :123-131) - 61 % of the forward FLOPs.
//
// conv3x3_v3 runs these with one MFMA wave per SIMD plus four loader waves that carry every byte HBM/L2 -> VGPR -> LDS;
// measured there: matrix pipe ~47 % busy, 440 vector-memory instructions and 380 KB of ds_write per 256-pixel tile, the
// epilogue exposed.  This kernel is built the other way round, after the shape the CDNA4 guide measures fastest for
// GEMM-like loops (8 waves, two per SIMD, LDS-DMA staging, counted vmcnt, raw barriers):
//   * 512 output pixels (16 x 32) per tile: the weights stream through LDS once per 512 pixels, not per 256;
//   * 8 MFMA waves, two per SIMD; wave w owns pixel rows 2w, 2w+1 x all COUT output channels (COUT accumulator registers):
//     COUT = 128: 6 fragment reads per 8 MFMAs, COUT = 64: 4 per 4; each SIMD always has a second wave to issue from while
//     one waits on LDS;
//   * all staging by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write, no loader waves.  The LDS
//     images are lane-linear, so the bank swizzle is applied to each lane's SOURCE address and again on the ds_read;
//   * K is walked as 4 chunks of 32 input channels (64 B per pixel) x 3 tap rows: one stage = 3 taps x 32 channels
//     (24 KB of weights for COUT = 128, ring of 3) against a double-buffered 18 x 34-pixel halo chunk (39 KB each);
//     ONE barrier per stage; stage s+2's weights and the next chunk's halo are in flight while stage s multiplies;
//   * fragment reads are hand-written ds_read_b128 with counted lgkmcnt waits, one per MFMA gap (hipcc stops counting LDS
//     waits in a kernel with LDS-DMA); the two waves of a SIMD (w, w+4) run their MFMAs one after the other, so waves 4-7
//     issue their DMAs before and waves 0-3 after their MFMAs; the next halo chunk's pieces go out over two stages;
//   * epilogue from the accumulators (bias pre-loaded into them, PReLU, v_permlane32_swap -> 64 contiguous bytes per lane,
//     residual, one bf16 rounding, 16-byte stores); the residual's first half is fetched before the tile's last stage;
//     before the stores lanes exchange pieces so that an instruction touches fewer lines (COUT = 64: quad transpose, whole
//     128-byte lines; COUT = 128: lane pairs, 32 contiguous bytes - all that fits beside 128 accumulators).
//   Measurements behind these choices: profiles/r01_final_inkernel_stamps.txt.
// RESM: 0 no residual | 2 the pair gather z (COUT = 128: t2 = z + PReLU(conv(t1))) | 3 s_i + alpha_partner * f into the
// view stack (COUT = 64, HRNet.py:123-131).
// LDS (COUT = 128): 3 x 24,576 (weights) + 2 x 39,936 (halo) + 512 (bias) = 154,112 B.
// Ordering rules followed (guide, "Pipelining across barriers"): a wave waits for its own DMAs with a counted vmcnt
// BEFORE the barrier that precedes the stage reading them; a buffer is re-filled only after a barrier that every reader
// of its previous contents has passed.
#include <type_traits>
#include "conv3x3.h"

namespace {

constexpr int T4_H = 16, T4_W = 32;
constexpr int HW4 = T4_W + 2;                             // halo width 34
constexpr int NPIX = (T4_H + 2) * HW4;                    // 612 halo pixels
constexpr int N_IN_DMA = (NPIX * 64 + 1023) / 1024;       // 39 wave-instructions of 1 KB per halo chunk
constexpr int IN_BYTES = N_IN_DMA * 1024;                 // 39,936

template <int COUT> struct V4Geo {
    static constexpr int NCB = COUT / 32;                 // 32-channel blocks per wave
    static constexpr int TAP_BYTES = COUT * 64;           // one tap x 32 cin
    static constexpr int WST_BYTES = 3 * TAP_BYTES;       // one stage
    static constexpr int W_PIECES = WST_BYTES / 1024;     // DMA wave-instructions per stage: 24 / 12
    static constexpr int OFF_IN = 3 * WST_BYTES;
    static constexpr int OFF_BIAS = OFF_IN + 2 * IN_BYTES;
    static constexpr int LDS_BYTES = OFF_BIAS + COUT * 4;
};

__device__ __attribute__((aligned(16))) unsigned hrn_v4_zero16[4];     // source of out-of-image halo pixels

__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ float raw_max4(float a, float b) {
    float y;
    asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
    return y;
}
__device__ __forceinline__ void wait_vm(int n) {          // counted wait; n is wave-uniform and one of the values below
    switch (n) {
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void lds_done_then_barrier4() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int COUT, int RESM>
__global__ __launch_bounds__(512, 2) void conv3x3_v4_kernel(const ConvParams p) {
    typedef V4Geo<COUT> GEO;
    constexpr int NCB = GEO::NCB, NPR = NCB / 2, WST_BYTES = GEO::WST_BYTES, TAP_BYTES = GEO::TAP_BYTES;
    constexpr int OFF_IN = GEO::OFF_IN, OPIX = COUT * 2;                   // bytes per output pixel
    constexpr bool RES = RESM != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bias_lds = (float*)(smem + GEO::OFF_BIAS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int H = p.H, W = p.W;
    const size_t hw = (size_t)H * W;
    const unsigned tiles_x = (W + T4_W - 1) / T4_W;
    const unsigned tiles_y = (H + T4_H - 1) / T4_H;
    const unsigned tiles = tiles_x * tiles_y;
    const unsigned total = tiles * (unsigned)p.M;
    const unsigned G = gridDim.x;
    const unsigned bid = blockIdx.x;
    const unsigned slot = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
    if (slot >= total) return;
    const int ntl = (int)((total - slot + G - 1) / G);
    unsigned cur_m = slot / tiles, cur_t = slot - cur_m * tiles;            // tile being multiplied
    const unsigned step_m = G / tiles, step_t = G - step_m * tiles;
    const bool in_pair = p.in_pair != 0;

    // view (image) base of input chunk c (32 channels = 64 B) of image m, and the pixel pitch there
    auto chunk_src = [&](unsigned m, int c, int& pitch) __attribute__((always_inline)) -> const unsigned char* {
        if (in_pair) {
            const int b = m / p.pair_h, i = m - b * p.pair_h;
            const int v = c < 2 ? i : p.pair_last - i;
            pitch = 128;
            return (const unsigned char*)p.stack + ((size_t)b * p.pair_vs + v) * hw * 128 + (c & 1) * 64;
        }
        pitch = 256;
        return (const unsigned char*)p.in + (size_t)m * hw * 256 + c * 64;
    };

    // ---- LDS-DMA issue: halo chunk c of tile (m, t) -> input buffer `buf`.  Wave w issues pieces j = w, w+8, ... < 39;
    // piece j, lane i -> LDS bytes j*1024 + i*16 = halo pixel j*16 + (i >> 2), physical 16-B chunk i & 3, which holds
    // logical chunk (i & 3) ^ ((pixel >> 2) & 3).
    const int n_in = w < (N_IN_DMA & 7) ? (N_IN_DMA >> 3) + 1 : (N_IN_DMA >> 3);
    auto issue_in = [&](unsigned m, unsigned t, int c, int buf, int jj_lo, int jj_hi) __attribute__((always_inline)) {
        const int ty = t / tiles_x;
        const int y0 = ty * T4_H, x0 = (t - ty * tiles_x) * T4_W;
        int pitch;
        const unsigned char* base = chunk_src(m, c, pitch);
        int lq = lane;
        asm volatile("" : "+v"(lq));                        // keep the per-piece geometry out of long-lived registers
#pragma unroll
        for (int jj = 0; jj < (N_IN_DMA + 7) / 8; ++jj) {
            const int j = w + 8 * jj;
            if (j < N_IN_DMA && jj >= jj_lo && jj < jj_hi) {
                const int pix = j * 16 + (lq >> 2);
                const int lc = (lq & 3) ^ ((pix >> 2) & 3);
                const int py = pix / HW4, px = pix - py * HW4;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool ok = pix < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                const unsigned char* src = ok ? base + ((size_t)(gy * W + gx) * pitch + lc * 16) : (const unsigned char*)hrn_v4_zero16;
                dma16(src, smem + OFF_IN + buf * IN_BYTES + j * 1024);
            }
        }
    };
    // ---- LDS-DMA issue: weights of stage (chunk c, tap row tg) -> ring slot: W_PIECES pieces of 16 couts x 64 B, piece q =
    // (tap kx = q / (COUT/16), cout group jj = q % (COUT/16)); wave w issues q = w, w+8, w+16 (< W_PIECES).
    // lane i -> cout 16jj + (i >> 2), physical chunk i & 3 = logical (i & 3) ^ ((cout >> 2) & 3).
    constexpr int GPT = COUT / 16;                          // pieces per tap
    const int n_w = (GEO::W_PIECES - w + 7) / 8;            // 3 (COUT 128) | 2 or 1 (COUT 64)
    const unsigned w_lane_off = (unsigned)((lane >> 2) * 128 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    auto issue_w = [&](int c, int tg, int slot_) __attribute__((always_inline)) {
        const unsigned char* base = (const unsigned char*)p.wpk + (size_t)((c >> 1) * 9 + tg * 3) * (COUT * 128) + (c & 1) * 64 + w_lane_off;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int q = w + 8 * t;
            if (q < GEO::W_PIECES) {
                const int kx = q / GPT, jj = q - kx * GPT;
                dma16(base + (size_t)kx * (COUT * 128) + jj * 2048, smem + slot_ * WST_BYTES + kx * TAP_BYTES + jj * 1024);
            }
        }
    };

    const bool has_slope = p.slope != nullptr;
    const float slope = has_slope ? p.slope[0] : 0.f;
    const bool slope01 = slope >= 0.f && slope <= 1.f;

    // fragment addresses.  Weights: cout row cb*32 + r, k-step ks -> a_off[ks] + cb*2048 + kx*TAP_BYTES + slot*WST_BYTES.
    // Input: halo pixel (2w + pb + tg, r + kx) -> b_off[pb + tg][kx] ^ (ks << 5) + buffer base (low bits zero).
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned a_off[2], a_off_hi[2], b_off[4][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { a_off[ks] = lds0 + (unsigned)(r * 64 + (((ks * 2 + hh) ^ ((r >> 2) & 3)) << 4)); a_off_hi[ks] = a_off[ks] + 32768; }
#pragma unroll
    for (int row = 0; row < 4; ++row)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int pix = (2 * w + row) * HW4 + r + kx;
            b_off[row][kx] = lds0 + (unsigned)(OFF_IN + pix * 64 + (((hh) ^ ((pix >> 2) & 3)) << 4));
        }

    f32x16 acc[NCB][2];                                     // [cout block][pixel row]

    if (tid < COUT) bias_lds[tid] = p.bias[tid];
    // prologue: weights of stages 0 and 1, halo chunk 0 of the first tile
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    issue_in(cur_m, cur_t, 0, 0, 0, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_done_then_barrier4();

    for (int tl = 0; tl < ntl; ++tl) {
        const bool more_tiles = tl + 1 < ntl;
        unsigned nxt_t = cur_t + step_t, nxt_m = cur_m + step_m;
        if (nxt_t >= tiles) { nxt_t -= tiles; ++nxt_m; }

        // the accumulators start at the bias (element 4g + j of block cb = channel cb*32 + 8g + 4hh + j)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *(const f32x4*)(bias_lds + cb * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[cb][0][4 * g + j] = b[j]; acc[cb][1][4 * g + j] = b[j]; }
            }

        for (int c = 0; c < 4; ++c) {
            const unsigned inbase = (unsigned)((c & 1) * IN_BYTES);
            auto stage = [&](auto tg_c) __attribute__((always_inline)) {
                constexpr int tg = decltype(tg_c)::value;
                // ---- in flight during this stage: weights of stage s+2 (ring slot (tg+2)%3), and at tg == 0 the next halo chunk.
                // The two waves of a SIMD (w, w+4) take the matrix pipe one after the other, so waves 4-7 issue their DMAs
                // before their MFMAs (while waves 0-3 multiply) and waves 0-3 after theirs (while waves 4-7 multiply).
                int issued = n_w;
                if (tg == 0 && (c < 3 || more_tiles)) issued += 3;               // halo pieces jj = 0..2 of the next chunk
                if (tg == 1 && (c < 3 || more_tiles)) issued += n_in - 3;        // and jj = 3.. (2, wave 7: 1)
                auto stage_issue = [&]() __attribute__((always_inline)) {
                    const int tg2 = (tg + 2) % 3, c2 = (c + (tg + 2) / 3) & 3;
                    issue_w(c2, tg2, tg2);
                    if (tg < 2) {                                   // the next halo chunk, spread over two stages
                        const int lo = tg == 0 ? 0 : 3, hi = tg == 0 ? 3 : 8;
                        if (c < 3) issue_in(cur_m, cur_t, c + 1, (c + 1) & 1, lo, hi);
                        else if (more_tiles) issue_in(nxt_m, nxt_t, 0, 0, lo, hi);
                    }
                };
                if (w >= 4) stage_issue();
                // ---- residual of the tile: pixel row 0's pieces are fetched before the tile's LAST stage multiplies, pixel row
                // 1's at the start of the epilogue, so that their HBM latency is covered by MFMA / epilogue work
                u32x4 rq[2][NPR][4];                            // [pixel row][cout pair][16-byte piece]
                auto res_fetch = [&](int pb) __attribute__((always_inline)) {
                    const int m = (int)cur_m;
                    const int ty = cur_t / tiles_x;
                    const int y0 = ty * T4_H, x0 = (cur_t - ty * tiles_x) * T4_W;
                    const int gy = y0 + 2 * w + pb, gx = x0 + r;
                    const int gyc = gy < H ? gy : H - 1, gxc = gx < W ? gx : W - 1;
#pragma unroll
                    for (int pr = 0; pr < NPR; ++pr) {
                        const unsigned char* view;
                        if (RESM == 2) {                        // z: channels 0..63 = view i, 64..127 = its partner
                            const int b = m / p.pair_h, i = m - b * p.pair_h;
                            view = (const unsigned char*)p.stack + ((size_t)b * p.pair_vs + (pr == 0 ? i : p.pair_last - i)) * hw * 128;
                        } else {                                // s_i of the view stack (the slot the output replaces)
                            const int b = m / p.out_h, i = m - b * p.out_h;
                            view = (const unsigned char*)p.res + ((size_t)b * p.res_vs + i) * hw * 128;
                        }
                        const u32x4* rp = (const u32x4*)(view + (unsigned)((gyc * W + gxc) * 128 + hh * 64));
#pragma unroll
                        for (int g = 0; g < 4; ++g) rq[pb][pr][g] = rp[g];
                    }
                };
                if (RES && c == 3 && tg == 2) res_fetch(0);
                // ---- 3 taps x 2 k-steps, fragment reads one step ahead of their MFMAs
                // Hand-issued fragment reads (see conv3x3_r64.hip: with LDS-DMA in the kernel hipcc answers every fragment use
                // with lgkmcnt(0)).  Reads of step i+1 go one per MFMA gap of step i, in the order A[0..NCB-1], B0, B1; the
                // MFMAs run (cb, pb) = (0,0) (0,1) (1,0) ...  LDS reads return in order: before (0,0) of step i, A[*](i) and
                // B0(i) are back once at most B1(i) is outstanding: lgkmcnt(1); before (0,1), B1(i): only A0(i+1), issued in
                // gap 0, is younger: lgkmcnt(1) (0 in the last step, which issues nothing).
                bf16x8 fa[2][NCB], fb[2][2];
                auto rd = [&](bf16x8& dst, unsigned addr, int imm) __attribute__((always_inline)) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm));
                };
                auto load_part = [&](int i, int part) __attribute__((always_inline)) {
                    const int s_ = i & 1, kx = i >> 1, ks = i & 1;
                    if (part < NCB) {
                        const int off = tg * WST_BYTES + kx * TAP_BYTES + part * 2048;
                        rd(fa[s_][part], off < 32768 ? a_off[ks] : a_off_hi[ks], off < 32768 ? off : off - 32768);
                    } else {
                        unsigned kbits = 0;
                        if (ks) asm volatile("s_mov_b32 %0, 32" : "=s"(kbits));
                        rd(fb[s_][part - NCB], (b_off[part - NCB + tg][kx] + inbase) ^ kbits, 0);
                    }
                };
#pragma unroll
                for (int part = 0; part < NCB + 2; ++part) load_part(0, part);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int s_ = i & 1;
                    const bool more = i + 1 < 6;
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
                            acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][cb], fb[s_][pb], acc[cb][pb], 0, 0, 0);
                            if (more && g < NCB + 2) load_part(i + 1, g);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
                if (w < 4) stage_issue();
                // everything issued before this stage has landed once only this stage's DMAs are outstanding:
                // stage s+1's weights (issued in stage s-1) and any halo chunk issued then
                wait_vm(issued);
                if (c == 3 && tg == 2) {
                    // ---- epilogue of this tile (registers + global memory only; overlaps nothing in LDS)
                    const int m = (int)cur_m;
                    const int ty = cur_t / tiles_x;
                    const int y0 = ty * T4_H, x0 = (cur_t - ty * tiles_x) * T4_W;
                    size_t oimg = (size_t)m;
                    float res_alpha = 1.f;
                    if (p.out_h > 0) {
                        const int ob = m / p.out_h, oi = m - ob * p.out_h;
                        oimg = (size_t)ob * p.out_vs + oi;
                        if (RESM == 3 && p.alphas) res_alpha = p.alphas[(size_t)ob * p.alpha_vs + (p.pair_last - oi)];
                    }
                    unsigned char* outp = (unsigned char*)p.out + (oimg * hw + (size_t)y0 * W + x0) * OPIX;
                    if (RES) res_fetch(1);                      // pixel row 1's residual: in flight while row 0 is finished
                    auto epilogue = [&](auto act_c) __attribute__((always_inline)) {
                        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
                        for (int pb = 0; pb < 2; ++pb) {
                            const int gy = y0 + 2 * w + pb;
#pragma unroll
                            for (int pr = 0; pr < NPR; ++pr) {          // cout blocks (2pr, 2pr+1) -> channels 64pr + 32hh ..
                                u32x4 uu[4];                             // COUT = 64: the group's four pieces | COUT = 128: two at a time
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    float xa[4], xb[4];
#pragma unroll
                                    for (int j = 0; j < 4; ++j) { xa[j] = acc[2 * pr][pb][4 * g + j]; xb[j] = acc[2 * pr + 1][pb][4 * g + j]; }
                                    if (ACT == 1) {
#pragma unroll
                                        for (int j = 0; j < 4; ++j) { xa[j] = raw_max4(xa[j], slope * xa[j]); xb[j] = raw_max4(xb[j], slope * xb[j]); }
                                    } else if (ACT == 2) {
#pragma unroll
                                        for (int j = 0; j < 4; ++j) { xa[j] = xa[j] >= 0.f ? xa[j] : slope * xa[j]; xb[j] = xb[j] >= 0.f ? xb[j] : slope * xb[j]; }
                                    }
                                    // v_permlane32_swap(a, b): lanes 32..63 of a <-> lanes 0..31 of b; afterwards a lane holds
                                    // (a, b) = channels 64pr + 32hh + 8g + (0..3, 4..7) of its pixel
                                    u32x4 u;
                                    if (RES) {
                                        float v[8];
#pragma unroll
                                        for (int j = 0; j < 4; ++j) {
                                            const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xa[j]), __float_as_uint(xb[j]), false, false);
                                            v[j] = __uint_as_float(sw[0]);
                                            v[4 + j] = __uint_as_float(sw[1]);
                                        }
#pragma unroll
                                        for (int j = 0; j < 4; ++j) {
                                            const float r0 = __uint_as_float(rq[pb][pr][g][j] << 16), r1 = __uint_as_float(rq[pb][pr][g][j] & 0xffff0000u);
                                            if (RESM == 3) { v[2 * j] = r0 + res_alpha * v[2 * j]; v[2 * j + 1] = r1 + res_alpha * v[2 * j + 1]; }
                                            else { v[2 * j] += r0; v[2 * j + 1] += r1; }
                                        }
#pragma unroll
                                        for (int j = 0; j < 4; ++j) u[j] = pack2_bf16(v[2 * j], v[2 * j + 1]);
                                    } else {
                                        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pack2_bf16(xa[0], xa[1]), pack2_bf16(xb[0], xb[1]), false, false);
                                        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pack2_bf16(xa[2], xa[3]), pack2_bf16(xb[2], xb[3]), false, false);
                                        u[0] = s0[0]; u[1] = s1[0]; u[2] = s0[1]; u[3] = s1[1];
                                    }
                                    if (COUT == 64) uu[g] = u;
                                    else {
                                        // COUT = 128 has no registers for the quad transpose: pieces are exchanged between the two
                                        // lanes of a pair instead (even lane: piece g of both pixels, odd lane: piece g+1), so that a
                                        // pair writes 32 contiguous bytes and an instruction touches 16 lines, not 32
                                        uu[g & 1] = u;
                                        if (g & 1) {
                                            const bool odd = (lane & 1) != 0;
                                            u32x4 s0, s1;
#pragma unroll
                                            for (int d = 0; d < 4; ++d) {
                                                const unsigned a = uu[0][d], b = uu[1][d];
                                                const unsigned ra = (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true);
                                                const unsigned rb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xF, 0xF, true);
                                                s0[d] = odd ? ra : a;           // even: own piece g-1 | odd: the even pixel's piece g
                                                s1[d] = odd ? b : rb;           // even: the odd pixel's piece g-1 | odd: own piece g
                                            }
                                            // even lane -> piece g-1, odd lane -> piece g; store 0 goes to the even pixel, store 1 to the odd one
                                            unsigned char* oe = outp + (unsigned)(((2 * w + pb) * W + (r & ~1)) * OPIX + pr * 128 + hh * 64 + (g - 1 + (lane & 1)) * 16);
                                            if (gy < H && x0 + (r & ~1) < W) *(u32x4*)oe = s0;
                                            if (gy < H && x0 + (r | 1) < W) *(u32x4*)(oe + OPIX) = s1;
                                        }
                                    }
                                }
                                if (COUT == 64) {   // registers to spare here: quad transpose -> every store instruction writes whole lines
                                    quad_transpose(uu, (lane & 1) != 0, (lane & 2) != 0);
                                    unsigned char* oq = outp + (unsigned)(((2 * w + pb) * W + (r & ~3)) * OPIX + pr * 128 + hh * 64 + (r & 3) * 16);
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
                                        if (gy < H && x0 + (r & ~3) + j < W) *(u32x4*)(oq + j * OPIX) = uu[j];
                                }
                            }
                        }
                    };
                    if (!has_slope) epilogue(std::integral_constant<int, 0>{});
                    else if (slope01) epilogue(std::integral_constant<int, 1>{});
                    else epilogue(std::integral_constant<int, 2>{});
                }
                lds_done_then_barrier4();
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            stage(std::integral_constant<int, 2>{});
        }
        cur_m = nxt_m; cur_t = nxt_t;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last stages' look-ahead DMAs must not outlive the workgroup
}

int g_v4_cus = 0;

template <int COUT, int RESM>
int launch_v4(const ConvParams& p, long grid, hipStream_t stream) {
    typedef V4Geo<COUT> GEO;
    static_assert(GEO::LDS_BYTES <= 160 * 1024, "LDS budget");
    { const int rc_lds = hrn_allow_lds((const void*)conv3x3_v4_kernel<COUT, RESM>, GEO::LDS_BYTES); if (rc_lds) return rc_lds; }
    hipLaunchKernelGGL((conv3x3_v4_kernel<COUT, RESM>), dim3((unsigned)grid), dim3(512), GEO::LDS_BYTES, stream, p);
    HRN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// bf16, 128 input channels.  COUT = 128: residual none or the pair gather (res_mode 2); COUT = 64: none or the alpha residual into
// the view stack (res_mode 3).  Returns -100 when not applicable.
int hrn_launch_conv3x3_v4(int cout, const ConvParams& p, hipStream_t stream) {
    if (p.scale || p.relu) return -100;
    if (cout == 128 && p.res_mode != 0 && p.res_mode != 2) return -100;
    if (cout == 64 && ((p.res_mode != 0 && p.res_mode != 3) || p.in_pair)) return -100;
    if (cout != 64 && cout != 128) return -100;
    if ((p.in_pair || p.res_mode == 2) && p.pair_h <= 0) return -100;
    if (p.res_mode == 3 && (p.out_h <= 0 || !p.res)) return -100;
    if (g_v4_cus == 0) {
        int dev = 0, n = 0;
        HRN_HIP(hipGetDevice(&dev));
        HRN_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        g_v4_cus = n > 0 ? n : 256;
    }
    const long tiles = (long)((p.W + T4_W - 1) / T4_W) * ((p.H + T4_H - 1) / T4_H);
    const long total = tiles * p.M;
    HRN_CHECK(total > 0, -2, "conv3x3_v4: bad tile count %ld", total);
    if (total >= (1L << 30) || (long)p.H * p.W * 256 >= (1L << 31)) return -100;     // 32-bit tile / in-image byte arithmetic
    long grid = g_v4_cus;
    if (total < grid) grid = total;
    if (grid >= 8) grid &= ~7L;
    const double px = (double)p.M * p.H * p.W;
    const char* fam = cout == 128 ? (p.res_mode ? "conv3x3_bf16_128x128+res" : "conv3x3_bf16_128x128")
                                  : (p.res_mode ? "conv3x3_bf16_128x64+res" : "conv3x3_bf16_128x64");
    HrnProfScope prof(fam, 2.0 * 128 * cout * 9 * px, px * 2 * (128 + cout + (p.res_mode ? cout : 0)), stream);
    if (cout == 128) return p.res_mode ? launch_v4<128, 2>(p, grid, stream) : launch_v4<128, 0>(p, grid, stream);
    return p.res_mode ? launch_v4<64, 3>(p, grid, stream) : launch_v4<64, 0>(p, grid, stream);
}
