// conv3x3 64 -> 64, bf16, RESIDENT weights + two ping-pong wave teams: the encoder's five 64->64 layers
// (HRNet.py:17-22, :55-60).
//
// Why a third kernel: 64->64 sits at the MFMA/HBM ridge (288 FLOP/B), and in conv3x3_v3 its time was neither: every
// 256-pixel tile re-staged all 72 KB of weights through the slow ds_write path (a barrier per 3 taps), and the epilogue
// (bias, PReLU, residual, rounding: ~2,000 VALU cycles per tile) ran while the matrix pipe idled.  For CIN = COUT = 64
// the whole weight tensor fits in LDS for the lifetime of a persistent workgroup, next to TWO halo tiles:
//     weights  9 x 64 x 128 B            73,728 B   16-byte chunks XOR-swizzled by (row >> 1) & 7: conflict-free b128 reads
//     input    2 x [10][34] px x 128 B   87,040 B   one halo tile per team, chunks swizzled by (pixel >> 1) & 7
//     bias                                   256 B                                        = 161,024 B
// 512 threads = two teams of four waves, one wave of each team on every SIMD.  The teams alternate:
//     ON  phase: 36 k-steps (9 taps x 4) on the team's halo tile - one uninterrupted hand-pipelined ds_read/MFMA stream,
//                no weight traffic, no address arithmetic beyond one v_xor per fragment, no barrier inside;
//                the tile's residual is fetched HBM -> VGPR at its start;
//     OFF phase: start the LDS-DMA (global_load_lds_dwordx4) of the team's next halo tile straight into the team's LDS
//                buffer - no staging registers, no ds_write; the bank swizzle is applied to each lane's SOURCE address -
//                then finish the tile just multiplied from the accumulators (bias pre-loaded, PReLU, v_permlane32_swap
//                so every lane owns 64 contiguous bytes of one pixel, residual, one bf16 rounding, whole-line stores),
//                then wait (counted vmcnt: the tile's own stores stay in flight) for the DMAs.
// One workgroup barrier per phase.  While team A's waves occupy the matrix pipe, team B's waves on the same SIMDs do the
// VALU / VMEM work, so neither the epilogue nor the input staging costs MFMA time.  No ordinary global load is issued in
// the OFF phase: beside an LDS-DMA in flight hipcc waits vmcnt(0) for any of them, which would drain the DMAs early.
// Same math / layouts / packed weights / persistent XCD-windowed tile walk as the other conv kernels.
// Measured in-kernel (profiles/r01_final_inkernel_stamps.txt): ON 5.4 k cycles for 4.6 k of matrix pipe since the fragment
// reads are hand-issued with counted lgkmcnt waits (multiply()); the OFF phase (6.3 k, 8.2 k with the residual) is bound by
// the CU's vector-memory pipe - 44 DMA + 32 store (+ 32 residual load) wave-instructions at ~60-70 cycles each.  Without
// residual, full tiles take their halo from precomputed lane offsets (issue(), fix_borders()).  Outputs leave - and the residual
// arrives - as whole 128-byte lines through a quad transpose, with the non-temporal policy: both are touched once per launch and
// read / written next by another launch from HBM anyway (round 2: the residual variant too; its own time +2 %, the launches
// around it -3 %).
#include <type_traits>
#include "conv3x3.h"

namespace {

constexpr int HALO_H = CONV_TILE_H + 2;
constexpr int HALO_W = CONV_TILE_W + 2;
constexpr int IN_BYTES = HALO_H * HALO_W * 128;              // 43,520 (unpadded, swizzled)
constexpr int W_BYTES = 9 * 64 * 128;                        // 73,728
constexpr int LDS_BYTES = W_BYTES + 2 * IN_BYTES + 256;

__device__ __forceinline__ void lds_done_then_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// v_max_f32 without the canonicalising v_max(x, x) the compiler puts in front of fmaxf (both operands NaN -> NaN, so
// PReLU still propagates NaNs)
__device__ __forceinline__ float raw_max(float a, float b) {
    float y;
    asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
    return y;
}

__device__ __attribute__((aligned(16))) unsigned hrn_r64_zero16[4];

template <bool RES>
__global__ __launch_bounds__(512, 2) void conv3x3_r64_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* w_lds = smem;
    float* bias_lds = (float*)(smem + W_BYTES + 2 * IN_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, tw = wave & 3;
    unsigned char* in_lds = smem + W_BYTES + team * IN_BYTES;
    const int r = lane & 31, hh = lane >> 5;
    const int H = p.H, W = p.W;
    const size_t hw = (size_t)H * W;
    const unsigned tiles_x = (W + CONV_TILE_W - 1) / CONV_TILE_W;
    const unsigned tiles_y = (H + CONV_TILE_H - 1) / CONV_TILE_H;
    const unsigned tiles = tiles_x * tiles_y;
    const unsigned total = tiles * (unsigned)p.M;           // < 2^31 (checked by the launcher)
    const unsigned G = gridDim.x;
    const unsigned bid = blockIdx.x;
    const unsigned slot = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
    if (slot >= total) return;
    const int ntl = (int)((total - slot + G - 1) / G);      // tiles of this workgroup; team T takes local tiles T, T+2, ...
    const int nmine = (ntl + 1 - team) >> 1;

    // this team's tile cursor (image, tile in image): tiles slot + (2k + team) * G, advanced without divisions
    unsigned cur_m = (slot + team * G) / tiles, cur_t = (slot + team * G) - cur_m * tiles;
    const unsigned step_m = (2 * G) / tiles, step_t = 2 * G - step_m * tiles;

    // ---- one-time: all nine weight slices -> LDS (chunk c of row `row` lives at chunk c ^ ((row >> 1) & 7)), bias
    {
        const u32x4* wg = (const u32x4*)p.wpk;                  // [tap][64 rows][8 chunks]
#pragma unroll
        for (int j = 0; j < W_BYTES / 16 / 512; ++j) {
            const int idx = tid + j * 512;
            const int row = idx >> 3, c = idx & 7;              // row = tap*64 + cout
            *(u32x4*)(w_lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = wg[idx];
        }
        if (tid < 64) bias_lds[tid] = p.bias[tid];
    }

    // ---- halo tile of (m, t): HBM -> this team's LDS buffer by LDS-DMA.  Wave tw of the team issues pieces j = tw, tw+4,
    // ... < 43; piece j, lane i -> LDS bytes j*1024 + i*16 = halo pixel j*8 + (i >> 3), physical 16-B chunk i & 7, which
    // holds logical chunk (i & 7) ^ ((pixel >> 1) & 7).  Out-of-image pixels are read from a zero line; the half-empty
    // last piece is EXEC-masked.  The geometry is recomputed from an opaque copy of the lane id so that it does not sit
    // in registers through the MFMA phase.
    // Residual-free variant: FULL tiles whose halo addresses all lie inside the tensor take a fast form - per-lane byte
    // offsets from the halo origin precomputed once (11 registers), no bounds tests, no zero-line select; pixels outside the
    // image are then zeroed in LDS (fix_borders) by the wave that fetched them, after its DMAs have landed.  The residual
    // variant has no registers to spare for the offsets (measured slower) and keeps the general form.
    unsigned voff[11], cm_lo = 0, cm_hi = 0;                // cm: 4 bits per piece: pixel in top row | bottom row | left | right column
    if (!RES) {
#pragma unroll
        for (int jj = 0; jj < 11; ++jj) {
            const int j = tw + 4 * jj;
            const int pix = j * 8 + (lane >> 3);
            const int py = pix / HALO_W, px = pix - py * HALO_W;
            voff[jj] = (unsigned)((py * W + px) * 128 + (((lane & 7) ^ ((pix >> 1) & 7)) << 4));
            unsigned cls = (py == 0 ? 1u : 0u) | (py == HALO_H - 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == HALO_W - 1 ? 8u : 0u);
            if (pix >= HALO_H * HALO_W) cls = 0;
            if (jj < 8) cm_lo |= cls << (4 * jj); else cm_hi |= cls << (4 * (jj - 8));
        }
    }
    unsigned pend_F = 0;                                    // border flags of the tile whose halo is in flight (fast form only)
    auto issue = [&](unsigned m, unsigned t) __attribute__((always_inline)) {
        const int ty = t / tiles_x;
        const int y0 = ty * CONV_TILE_H, x0 = (t - ty * tiles_x) * CONV_TILE_W;
        const unsigned char* base = (const unsigned char*)p.in + (size_t)m * hw * 128;     // uniform: image base
        const bool fast = !RES && y0 + CONV_TILE_H <= H && x0 + CONV_TILE_W <= W && (m > 0 || y0 > 0) &&
                          ((int)m + 1 < p.M || (size_t)(y0 + CONV_TILE_H) * W + x0 + CONV_TILE_W < hw);
        pend_F = 0;
        if (fast) {
            const unsigned char* hb = base + ((long)(y0 - 1) * W + (x0 - 1)) * 128;
#pragma unroll
            for (int jj = 0; jj < 11; ++jj) {
                const int j = tw + 4 * jj;
                if (j < 43) {
                    if (jj < 10 || j * 8 + (lane >> 3) < HALO_H * HALO_W)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(hb + voff[jj]),
                                                         (__attribute__((address_space(3))) void*)(in_lds + j * 1024), 16, 0, 0);
                }
            }
            pend_F = (y0 == 0 ? 1u : 0u) | (y0 + CONV_TILE_H == H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + CONV_TILE_W == W ? 8u : 0u);
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        int lq = lane;
        asm volatile("" : "+v"(lq));
#pragma unroll
        for (int jj = 0; jj < 11; ++jj) {
            const int j = tw + 4 * jj;
            if (j < 43) {
                const int pix = j * 8 + (lq >> 3);
                const int lc = (lq & 7) ^ ((pix >> 1) & 7);
                const int py = pix / HALO_W, px = pix - py * HALO_W;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                const unsigned char* src = ok ? base + (unsigned)((gy * W + gx) * 128 + lc * 16) : (const unsigned char*)hrn_r64_zero16;
                if (pix < HALO_H * HALO_W)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(in_lds + j * 1024), 16, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto fix_borders = [&]() __attribute__((always_inline)) {
        if (!RES && pend_F) {
            const unsigned fm = pend_F * 0x11111111u;
            const unsigned hit_lo = cm_lo & fm, hit_hi = cm_hi & fm;
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int jj = 0; jj < 11; ++jj) {
                const int j = tw + 4 * jj;
                if (j < 43) {
                    const unsigned h = jj < 8 ? (hit_lo >> (4 * jj)) & 15u : (hit_hi >> (4 * (jj - 8))) & 15u;
                    if (h) *(u32x4*)(in_lds + j * 1024 + lane * 16) = z;
                }
            }
        }
    };

    const bool has_slope = p.slope != nullptr;
    const float slope = has_slope ? p.slope[0] : 0.f;
    const bool slope01 = slope >= 0.f && slope <= 1.f;
    // fragment addresses.  Weights: row r of cout block cb, k-step ks -> a_off[ks] + cb*4096 + tap*8192.
    // Input: halo pixel (2*tw + pb + ky, r + kx), k-step ks -> b_off[pb + ky][kx] ^ (ks << 5)   (12 registers, not 72;
    // the buffer base has zero low bits, so it is folded in before the xor)
    // All of them are absolute LDS byte addresses (the fragment reads are hand-written ds_read_b128, see multiply()); the
    // 16-bit instruction offset reaches taps 0-3 from a_off and taps 4-8 from a_off_hi.
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned a_off[4], a_off_hi[4], b_off[4][3];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a_off[ks] = lds0 + r * 128 + (((ks * 2 + hh) ^ ((r >> 1) & 7)) << 4);
        a_off_hi[ks] = a_off[ks] + 4 * 8192;
    }
#pragma unroll
    for (int row = 0; row < 4; ++row)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int pix = (2 * tw + row) * HALO_W + r + kx;
            b_off[row][kx] = lds0 + (unsigned)(W_BYTES + team * IN_BYTES) + (((unsigned)pix << 7) | ((unsigned)(((pix >> 1) & 7) ^ hh) << 4));
        }

    f32x16 acc[2][2];                                       // [cout block][pixel row]; lives from the ON phase into the OFF phase

    // ---- ON phase: the 36 k-steps of one tile
    auto multiply = [&]() __attribute__((always_inline)) {
        // the accumulators start at the bias (element 4g + j of block cb = channel cb*32 + 8g + 4hh + j)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *(const f32x4*)(bias_lds + cb * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[cb][0][4 * g + j] = b[j]; acc[cb][1][4 * g + j] = b[j]; }
            }
        // Hand-issued fragment stream.  In a kernel that also issues LDS-DMA hipcc does not count LDS waits: it answers
        // every fragment use with s_waitcnt lgkmcnt(0), which waits for the reads issued a moment ago as well and exposed
        // the whole LDS latency every third k-step (in-kernel stamps: 6,800 cycles per tile for 4,608 cycles of MFMA).  So
        // the ds_read_b128 are inline asm - invisible to the compiler's wait insertion - and are waited for with counted
        // lgkmcnt here; and they are spread one per MFMA gap instead of four in a burst in front of the four MFMAs (all
        // four waves of the team burst together, and an MFMA cannot issue before the reads in front of it have).
        //   gap 0: A0(i+2)   gap 1: A1(i+2)   gap 2: B0(i+2)   gap 3: B1(i+2)     [A = weights, B = pixels, two steps ahead]
        // MFMA order (cout block, pixel row) = (0,0) (0,1) (1,0) (1,1).  LDS reads return in order, so before (0,0) of step
        // i the reads A0(i), A1(i), B0(i) are back once at most the five younger ones - B1(i) and the four of step i+1 -
        // are outstanding: lgkmcnt(5); before (0,1), B1(i): the four of step i+1 and A0(i+2): lgkmcnt(5) again.  The last
        // two steps issue nothing and count down.  The asm statements are volatile: they keep their order, and each wait
        // names the fragments it guards as in/out operands, so no MFMA can move in front of its wait.  The reads of one
        // gap may be scheduled in front of that gap's MFMA but never across the sched_barrier to another gap's wait.
        constexpr int NK = 36;
        bf16x8 fa[3][2], fb[3][2];
        auto rd = [&](bf16x8& dst, unsigned addr, int imm) __attribute__((always_inline)) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm));
        };
        auto load_part = [&](int i, int part) __attribute__((always_inline)) {
            const int s_ = i % 3, tap = i >> 2, ks = i & 3;
            const int ky = tap / 3, kx = tap - ky * 3;
            if (part < 2) {
                rd(fa[s_][part], tap < 4 ? a_off[ks] : a_off_hi[ks], (tap < 4 ? tap : tap - 4) * 8192 + part * 4096);
            } else {
                unsigned kbits = 0;
                if (ks) asm volatile("s_mov_b32 %0, %1" : "=s"(kbits) : "n"(ks << 5));     // opaque: keeps the compiler from
                rd(fb[s_][part - 2], b_off[part - 2 + ky][kx] ^ kbits, 0);                  // materialising all 72 addresses
            }
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // nothing of the compiler's own left in the LGKM queue
        __builtin_amdgcn_s_setprio(3);                              // the matrix-pipe stream wins issue ties against the OFF team
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int part = 0; part < 4; ++part) load_part(i, part);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int s_ = i % 3;
            const bool more = i + 2 < NK;
            if (i < NK - 1) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa[s_][0]), "+v"(fb[s_][0]));
            else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa[s_][0]), "+v"(fb[s_][0]));
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][0], fb[s_][0], acc[0][0], 0, 0, 0);
            if (more) load_part(i + 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fb[s_][1]));
            else if (i == NK - 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fb[s_][1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[s_][1]));
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][0], fb[s_][1], acc[0][1], 0, 0, 0);
            if (more) load_part(i + 2, 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(fa[s_][1]));                     // A1(i) is older than B0(i): already back
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][1], fb[s_][0], acc[1][0], 0, 0, 0);
            if (more) load_part(i + 2, 2);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][1], fb[s_][1], acc[1][1], 0, 0, 0);
            if (more) load_part(i + 2, 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- residual of the tile at the cursor: the 64 bytes this lane will own after the swap (channels 32*hh .. 32*hh+31
    // of its pixel).  Fetched at the START of the ON phase, consumed in the OFF phase - a whole K loop of latency cover.
    u32x4 resv[2][4];
    float res_alpha = 1.f;
    auto fetch_residual = [&]() __attribute__((always_inline)) {
        const int m = (int)cur_m;
        const int ty = cur_t / tiles_x;
        const int y0 = ty * CONV_TILE_H, x0 = (cur_t - ty * tiles_x) * CONV_TILE_W;
        const unsigned char* rbase = (const unsigned char*)p.res + (size_t)m * hw * 128;               // res_mode 1 (image base)
        if (p.res_mode == 3) {
            const int ob = m / p.out_h, oi = m - ob * p.out_h;
            rbase = (const unsigned char*)p.res + ((size_t)ob * p.res_vs + oi) * hw * 128;
            res_alpha = p.alphas ? p.alphas[(size_t)ob * p.alpha_vs + (p.pair_last - oi)] : 1.f;
        }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int gy = y0 + 2 * tw + pb, gyc = gy < H ? gy : H - 1;
            // whole 128-byte lines per instruction (lane a of a quad: piece a of the quad's four pixels), transposed back in the
            // epilogue; read once: non-temporal (with 16-byte pieces per instruction that costs +16 %: nothing merges them)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gxj = x0 + (r & ~3) + j, gxjc = gxj < W ? gxj : W - 1;
                resv[pb][j] = __builtin_nontemporal_load((const u32x4*)(rbase + (unsigned)((gyc * W + gxjc) * 128 + hh * 64 + (r & 3) * 16)));
            }
        }
    };

    // ---- OFF phase: finish the tile at the cursor from the accumulators, refill the team's halo buffer with the next one
    auto finish = [&](bool more) __attribute__((always_inline)) {
        const int m = (int)cur_m;
        const int ty = cur_t / tiles_x;
        const int y0 = ty * CONV_TILE_H, x0 = (cur_t - ty * tiles_x) * CONV_TILE_W;
        cur_t += step_t; cur_m += step_m;
        if (cur_t >= tiles) { cur_t -= tiles; ++cur_m; }
        size_t oimg = (size_t)m;
        int ob = 0, oi = 0;
        if (p.out_h > 0) { ob = m / p.out_h; oi = m - ob * p.out_h; oimg = (size_t)ob * p.out_vs + oi; }
        // uniform base at the tile origin; per-lane byte offset of (row 2*tw, column r, channel half hh) is tile-independent
        unsigned char* outp = (unsigned char*)p.out + (oimg * hw + (size_t)y0 * W + x0) * 128;
        if (RES) {                                          // make the residual's wait happen BEFORE the DMAs are issued
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int g = 0; g < 4; ++g) asm volatile("" :: "v"(resv[pb][g]));
            asm volatile("" :: "v"(res_alpha));
        }
        if (more) issue(cur_m, cur_t);                      // in flight while the epilogue runs
        // ACT 0: no activation | 1: PReLU with 0 <= slope <= 1 as max(x, slope*x) (2 VALU) | 2: general PReLU (3 VALU)
        auto epilogue = [&](auto act_c) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int gy = y0 + 2 * tw + pb;
            u32x4 uu[4];                                     // the row's four pieces, stored together (whole-line stores)
            if (RES) quad_transpose(resv[pb], (lane & 1) != 0, (lane & 2) != 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float xa[4], xb[4];                         // channels 8g + 4hh + j of cout block 0 / block 1
#pragma unroll
                for (int j = 0; j < 4; ++j) { xa[j] = acc[0][pb][4 * g + j]; xb[j] = acc[1][pb][4 * g + j]; }
                if (ACT == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xa[j] = raw_max(xa[j], slope * xa[j]); xb[j] = raw_max(xb[j], slope * xb[j]); }
                } else if (ACT == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xa[j] = xa[j] >= 0.f ? xa[j] : slope * xa[j]; xb[j] = xb[j] >= 0.f ? xb[j] : slope * xb[j]; }
                }
                // v_permlane32_swap(a, b): lanes 32..63 of a <-> lanes 0..31 of b.  Afterwards a lane holds, for its pixel,
                // (a, b) = channels 32*hh + 8g + (0..3, 4..7): 16 contiguous bytes once rounded to bf16.
                u32x4 u;
                if (RES) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xa[j]), __float_as_uint(xb[j]), false, false);
                        v[j] = __uint_as_float(sw[0]);
                        v[4 + j] = __uint_as_float(sw[1]);
                    }
                    const u32x4 rq = resv[pb][g];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float r0 = __uint_as_float(rq[j] << 16), r1 = __uint_as_float(rq[j] & 0xffff0000u);
                        if (p.res_mode == 3) { v[2 * j] = r0 + res_alpha * v[2 * j]; v[2 * j + 1] = r1 + res_alpha * v[2 * j + 1]; }
                        else { v[2 * j] += r0; v[2 * j + 1] += r1; }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = pack2_bf16(v[2 * j], v[2 * j + 1]);
                } else {                                    // no residual: round first, swap packed pairs (half the swaps)
                    const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pack2_bf16(xa[0], xa[1]), pack2_bf16(xb[0], xb[1]), false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pack2_bf16(xa[2], xa[3]), pack2_bf16(xb[2], xb[3]), false, false);
                    u[0] = s0[0]; u[1] = s1[0]; u[2] = s0[1]; u[3] = s1[1];
                }
                uu[g] = u;
            }
            {   // whole-line stores: the four lanes of a quad exchange their pieces so that each instruction writes whole 128-byte lines
                quad_transpose(uu, (lane & 1) != 0, (lane & 2) != 0);
                unsigned char* oq = outp + (unsigned)(((2 * tw + pb) * W + (r & ~3)) * 128 + hh * 64 + (r & 3) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    // read next by another launch, from HBM anyway: non-temporal.  A/B on one box, both variants with whole-line nt
                    // accesses against plain stores / 16-byte residual pieces: 64->64 1.015 -> 0.986 ms, + residual 1.328 -> 1.356,
                    // the encoder -0.03 ms and the launches after it a little faster (less of the L2 turned over)
                    if (gy < H && x0 + (r & ~3) + j < W) __builtin_nontemporal_store(uu[j], (u32x4*)(oq + j * 128));
            }
        }
        };
        if (!has_slope) epilogue(std::integral_constant<int, 0>{});
        else if (slope01) epilogue(std::integral_constant<int, 1>{});
        else epilogue(std::integral_constant<int, 2>{});
        {   // the halo DMAs were issued before this tile's stores: wait until only those stores are outstanding
            const int nst = 4 * ((y0 + 2 * tw < H) + (y0 + 2 * tw + 1 < H));
            if (nst == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (nst == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        fix_borders();
    };

    if (nmine > 0) issue(cur_m, cur_t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fix_borders();
    lds_done_then_barrier();                                // weights, bias, both teams' first halo tiles

    // phase ph: team 0 is at step q = ph, team 1 at q = ph - 1; even q = ON (tile q/2), odd q = OFF (tile (q-1)/2)
    for (int ph = 0; ph <= ntl; ++ph) {
        const int q = ph - team;
        if (q >= 0) {
            const int k = q >> 1;
            if (k < nmine) {
                if ((q & 1) == 0) { if (RES) fetch_residual(); multiply(); }
                else finish(k + 1 < nmine);
            }
        }
        if (ph < ntl) lds_done_then_barrier();
    }
}


template <bool RES>
int launch_r64(const ConvParams& p, long grid, hipStream_t stream) {
    { const int rc_lds = hrn_allow_lds((const void*)conv3x3_r64_kernel<RES>, LDS_BYTES); if (rc_lds) return rc_lds; }
    hipLaunchKernelGGL(conv3x3_r64_kernel<RES>, dim3((unsigned)grid), dim3(512), LDS_BYTES, stream, p);
    HRN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// bf16 64 -> 64 with a plain input tensor (no pair gather).  Returns -100 when not applicable.
int hrn_launch_conv3x3_r64(const ConvParams& p, hipStream_t stream) {
    if (p.scale || p.relu || p.in_pair || p.res_mode == 2) return -100;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    const long tiles = (long)((p.W + CONV_TILE_W - 1) / CONV_TILE_W) * ((p.H + CONV_TILE_H - 1) / CONV_TILE_H);
    const long total = tiles * p.M;
    HRN_CHECK(total > 0, -2, "conv3x3_r64: bad tile count %ld", total);
    if (total >= (1L << 30) || (long)p.H * p.W * 128 >= (1L << 31)) return -100;     // 32-bit tile / in-image byte arithmetic
    long grid = hrn_device_cus();
    if (total < grid) grid = total;
    if (grid >= 8) grid &= ~7L;
    const double px = (double)p.M * p.H * p.W;
    HrnProfScope prof(p.res_mode ? "conv3x3_bf16_64x64+res" : "conv3x3_bf16_64x64", 2.0 * 64 * 64 * 9 * px,
                      px * 2 * (64 + 64 + (p.res_mode ? 64 : 0)), stream);
    return p.res_mode ? launch_r64<true>(p, grid, stream) : launch_r64<false>(p, grid, stream);
}
