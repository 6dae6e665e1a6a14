// conv3x3 bf16, wave-specialised persistent kernel for gfx950 (v3).
//
// Same math, layouts and packed weights as conv3x3.hip (D[cout][pixel] = W[cout][k] X[k][pixel]; NHWC bf16;
// weights [step = chunk*9 + tap][COUT][64 k]).  What changes is WHO does what.  Findings that drove it (r01 profiles
// and ablations, profiles/r01_*):
//   * with every wave staging AND computing, 61 % of wave cycles were s_waitcnt / s_barrier, the MFMA pipe 27-32 % busy;
//   * vmcnt retires in order: a wave that waits for its small per-step weight load also waits for every older memory
//     operation of its own - the previous tile's HBM stores, the next tile's HBM prefetch - so "asynchronous" traffic
//     was serialised at the next weight wait;
//   * one workgroup per tile cost 0.3-0.75 ms per layer in dispatch alone;
//   * hipcc re-uses one fragment register set and waits on every ds_read it has just issued: a hand-pipelined
//     fragment ring lifts a lone compute wave per SIMD to 71-78 % of the MFMA peak (compute-only ablation);
//   * 8-byte-per-lane NHWC stores / residual loads straight from the accumulator layout touch 32 partial 128-byte
//     lines per wave instruction; going through an LDS transpose makes every access a run of full lines.
// One persistent 512-thread workgroup per CU, 8 waves = 2 per SIMD, each SIMD pairing a matrix wave with a memory wave
// (MFMA and VALU/VMEM issue from different waves co-execute):
//   waves 0-3  COMPUTE : ds_read_b128 fragments (ring, 1-2 k-steps ahead) + MFMA, never a vmcnt wait inside the K loop;
//                        at the end of a tile the epilogue (bias, PReLU, LDS transpose, residual, full-line stores).
//   waves 4-5  WEIGHTS : stream the weight stages L2 -> VGPR -> LDS, two stages in flight (own vmcnt queue).
//   waves 6-7  INPUT   : fetch the NEXT (tile, channel-chunk) halo tile HBM -> VGPR -> the other LDS input buffer while
//                        the current one is multiplied (own vmcnt queue).
// Hand-offs are raw s_barrier (no compiler fence draining loads in flight), one per weight stage:
//   COUT = 64 : stage = 3 taps (48 MFMA per compute wave between barriers), COUT = 128: stage = 1 tap (32 MFMA).
// LDS: 2 input halo tiles [10][34] px x 144 B, 2 weight stages, bias = 153.5 KB (COUT 64) / 135.3 KB (COUT 128).
// Known cost left on the table (r01 ablations, profiles/r01_v3_*.txt): an epilogue run by the lone compute wave of each
// SIMD leaves the MFMA pipe idle (+0.4 / +0.75 / +0.23 ms per 128x128 / 64x64 / 128x64 launch at c3), and the per-tile
// weight re-staging keeps the LDS 60-90 % busy (compute waves alone: 0.53 ms, with the loaders: 0.79 ms for 128x128).
// The OFFLOAD variants below move the stores to the INPUT waves (they gain 2-8 %); variants in which a memory wave also
// had to LOAD for the epilogue (residual at store time, or a 16-piece residual ring), a tile-level software pipeline
// with two accumulator sets, and delaying the loaders' LDS writes (s_sleep) were all correct but not faster - any
// wave that is late for a stage barrier stalls everybody - and are not kept.
#include "conv3x3.h"
#include <stdlib.h>

namespace {

constexpr int HALO_H = CONV_TILE_H + 2;
constexpr int HALO_W = CONV_TILE_W + 2;
constexpr int PIX_PITCH = 144;
constexpr int IN_BUF_BYTES = HALO_H * HALO_W * PIX_PITCH;   // 48,960
constexpr int W_ROW_PITCH = 144;
constexpr int N_IN_PIECES = HALO_H * HALO_W * 8;            // 2,720 16-byte pieces
constexpr int LOADER_THREADS = 128;                          // per role

__device__ __forceinline__ void lds_done_then_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my ds_writes have landed / my ds_reads have returned
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct V3Tile {
    int m, y0, x0;
    const unsigned char* src0;
    const unsigned char* src1;
    const unsigned char* in_plain;
};

constexpr int TILE_PX = CONV_TILE_H * CONV_TILE_W;          // 256

template <int COUT, bool OFFLOAD> struct V3Geom {
    static constexpr int TPS = COUT == 64 ? 3 : 1;                      // taps per weight stage
    static constexpr int W_BUF_BYTES = TPS * COUT * W_ROW_PITCH;
    // OFFLOAD: bf16 staging tile [256 px][COUT] (+16 B pad per pixel) in the input buffer a finished tile releases;
    // rows that do not fit it (COUT = 128: 76 of 256) live in a spare region behind the bias
    static constexpr int SP = COUT * 2 + 16;
    static constexpr int SROWS_IN = IN_BUF_BYTES / SP < TILE_PX ? IN_BUF_BYTES / SP : TILE_PX;
    static constexpr int SPARE_BYTES = OFFLOAD ? (TILE_PX - SROWS_IN) * SP : 0;
    static constexpr int LDS_BYTES = 2 * IN_BUF_BYTES + 2 * W_BUF_BYTES + COUT * 4 + SPARE_BYTES;
};

// OFFLOAD = true: the compute waves end a tile with bias + PReLU (+ residual, COUT = 64) + ONE bf16 rounding into an LDS
// staging tile, and the INPUT waves - which never load anything for it, so they are never late for a stage barrier -
// copy it out as 16-byte-per-lane, full-128-byte-line stores while the next tile is already being multiplied.
// OFFLOAD = false: the compute waves run the whole epilogue themselves (used for COUT = 128 layers with a residual,
// whose 64 residual registers per lane do not fit next to 128 accumulators).
template <int CIN, int COUT, bool OFFLOAD>
__global__ __launch_bounds__(512, 2) void conv3x3_v3_kernel(const ConvParams p) {
    typedef V3Geom<COUT, OFFLOAD> GEO;
    constexpr int ES = 2;
    constexpr int NCHUNK = CIN / 64;
    constexpr int NCB = COUT / 32;                          // cout blocks per compute wave (all of COUT)
    constexpr int TPS = GEO::TPS;
    constexpr int NST = 9 / TPS;                            // stages per (tile, chunk) phase
    constexpr int NST_TILE = NST * NCHUNK;                  // stages per tile
    constexpr int W_BUF_BYTES = GEO::W_BUF_BYTES;
    constexpr int W_PIECES = TPS * COUT * 8;
    constexpr int PW = W_PIECES / LOADER_THREADS;           // weight pieces per loader thread and stage (12 / 8)
    constexpr int PI = (N_IN_PIECES + LOADER_THREADS - 1) / LOADER_THREADS;    // 22
    static_assert(W_PIECES % LOADER_THREADS == 0, "weight stage must split evenly");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* in_lds = smem;                                   // [2][IN_BUF_BYTES]
    unsigned char* w_lds = smem + 2 * IN_BUF_BYTES;                 // [2][W_BUF_BYTES]
    float* bias_lds = (float*)(w_lds + 2 * W_BUF_BYTES);            // [COUT]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W;
    const size_t hw = (size_t)H * W;
    const int tiles_x = (W + CONV_TILE_W - 1) / CONV_TILE_W;
    const int tiles_y = (H + CONV_TILE_H - 1) / CONV_TILE_H;
    const int tiles = tiles_x * tiles_y;
    const long total = (long)tiles * p.M;

    const int G = gridDim.x;
    const int bid = blockIdx.x;
    const int slot = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
    if (slot >= total) return;
    const int ntl = (int)((total - slot + G - 1) / G);      // tiles this workgroup walks: slot, slot+G, ...
    const long nstage_total = (long)ntl * NST_TILE;

    const bool in_pair = p.in_pair != 0;
    const int in_pix_bytes = in_pair ? 128 : CIN * ES;

    auto make_tile = [&](long tl) {
        V3Tile c;
        c.m = (int)(tl / tiles);
        const int t = (int)(tl - (long)c.m * tiles);
        const int ty = t / tiles_x;
        c.y0 = ty * CONV_TILE_H;
        c.x0 = (t - ty * tiles_x) * CONV_TILE_W;
        c.src0 = c.src1 = nullptr;
        if (p.pair_h > 0) {
            const int b = c.m / p.pair_h, i = c.m - b * p.pair_h;
            c.src0 = (const unsigned char*)p.stack + ((size_t)b * p.pair_vs + i) * hw * 128;
            c.src1 = (const unsigned char*)p.stack + ((size_t)b * p.pair_vs + (p.pair_last - i)) * hw * 128;
        }
        c.in_plain = (const unsigned char*)p.in + (size_t)c.m * hw * CIN * ES;
        return c;
    };
    // OFFLOAD staging: pixel row of the tile -> LDS address (rows beyond SROWS_IN live in the spare region)
    constexpr int SP = GEO::SP;
    constexpr int SROWS_IN = GEO::SROWS_IN;
    unsigned char* spare_lds = (unsigned char*)(bias_lds + COUT);
    auto stage_row = [&](unsigned char* stg0, int row) __attribute__((always_inline)) -> unsigned char* {
        if constexpr (SROWS_IN >= TILE_PX) return stg0 + row * SP;
        else return row < SROWS_IN ? stg0 + row * SP : spare_lds + (row - SROWS_IN) * SP;
    };
    (void)stage_row; (void)spare_lds;

    if (tid < COUT) bias_lds[tid] = p.bias[tid];

    if (wave >= 6) {
        // ================================================================ INPUT loader + STORE duty (waves 6,7)
        const int lt = tid - 384;
        u32x4 reg[PI];
        auto issue = [&](long phase) __attribute__((always_inline)) {      // phase = local tile index * NCHUNK + chunk
            const long tl = slot + (phase / NCHUNK) * G;
            const int chunk = (int)(phase % NCHUNK);
            const V3Tile c = make_tile(tl);
            const unsigned char* base = in_pair ? (chunk == 0 ? c.src0 : c.src1) : c.in_plain + chunk * 128;
#pragma unroll
            for (int it = 0; it < PI; ++it) {
                const int cc = lt + it * LOADER_THREADS;
                const int pix = cc >> 3, part = cc & 7;
                const int py = pix / HALO_W, px = pix - py * HALO_W;
                const int gy = c.y0 + py - 1, gx = c.x0 + px - 1;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (cc < N_IN_PIECES && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                    v = *(const u32x4*)(base + ((size_t)gy * W + gx) * in_pix_bytes + part * 16);
                reg[it] = v;
            }
        };
        auto commit = [&](int buf) __attribute__((always_inline)) {
            unsigned char* dst = in_lds + buf * IN_BUF_BYTES;
#pragma unroll
            for (int it = 0; it < PI; ++it) {
                const int cc = lt + it * LOADER_THREADS;
                if (cc < N_IN_PIECES) *(u32x4*)(dst + (cc >> 3) * PIX_PITCH + (cc & 7) * 16) = reg[it];
            }
        };
        // ---- OFFLOAD store duty.  A finished tile's staging tile is complete after the FIRST barrier of the next phase and
        // has to be gone before this wave's own input commit at the end of that phase (same buffer).  Its NSP 16-byte
        // pieces per thread are spread over the NSLOT = NST-1 stages in between; a piece is one ds_read_b128 and one
        // global store (8 lanes per 128-byte pixel line) - no loads, so this wave is never late for a stage barrier.
        constexpr int PPP = COUT / 8;                           // 16-byte pieces per staged pixel
        constexpr int NSP = TILE_PX * PPP / LOADER_THREADS;     // pieces per thread and tile (16 / 32)
        constexpr int NSLOT = NST - 1;
        constexpr int QS = (NSP + NSLOT - 1) / NSLOT;           // pieces per slot
        struct Duty { int y0, x0; unsigned char* outp; };
        auto make_duty = [&](int tli) __attribute__((always_inline)) {
            Duty d;
            const V3Tile c = make_tile(slot + (long)tli * G);
            size_t out_img = (size_t)c.m;
            if (p.out_h > 0) { const int b = c.m / p.out_h, i = c.m - b * p.out_h; out_img = (size_t)b * p.out_vs + i; }
            d.outp = (unsigned char*)p.out + out_img * hw * COUT * ES;
            d.y0 = c.y0; d.x0 = c.x0;
            return d;
        };
        auto store_slot = [&](const Duty& d, unsigned char* stg0, int k) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < QS; ++i) {
                const int j = k * QS + i;
                if (j < NSP) {
                    const int idx = lt + j * LOADER_THREADS;
                    const int pixel = idx / PPP, part = idx - pixel * PPP;
                    const int gy = d.y0 + (pixel >> 5), gx = d.x0 + (pixel & 31);
                    const u32x4 v = *(const u32x4*)(stage_row(stg0, pixel) + part * 16);
                    if (gy < H && gx < W) *(u32x4*)(d.outp + (((size_t)gy * W + gx) * COUT + part * 8) * ES) = v;
                }
            }
        };
        const long nphase = (long)ntl * NCHUNK;
        issue(0);
        commit(0);
        lds_done_then_barrier();                            // prologue barrier
        for (long ph = 0; ph < nphase; ++ph) {
            const bool more = ph + 1 < nphase;
            if (more) issue(ph + 1);                        // in flight during the whole phase
            const bool duty = OFFLOAD && ph > 0 && ph % NCHUNK == 0;   // the tile that ended with phase ph-1 is drained now
            Duty d = {0, 0, nullptr};
            if (duty) d = make_duty((int)(ph / NCHUNK) - 1);
            unsigned char* stg0 = in_lds + (int)((ph - 1) & 1) * IN_BUF_BYTES;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (st == NST - 1 && more) commit((int)((ph + 1) & 1));     // buffer released at the end of phase ph-1
                lds_done_then_barrier();
                if constexpr (OFFLOAD) {
                    if (st < NSLOT && duty) store_slot(d, stg0, st);
                }
            }
        }
        if constexpr (OFFLOAD) {
            lds_done_then_barrier();                        // final barrier: the last tile is staged
            const Duty d = make_duty(ntl - 1);
            unsigned char* stg0 = in_lds + (int)((nphase - 1) & 1) * IN_BUF_BYTES;
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) store_slot(d, stg0, k);
        }
        return;
    }

    if (wave >= 4) {
        // ================================================================ WEIGHT loader (waves 4,5)
        const int lt = tid - 256;
        // (native vector type on purpose: an array of HIP's uint4 STRUCT carried around a loop is demoted to scratch)
        const u32x4* wg = (const u32x4*)p.wpk;
        const int dst0 = (lt >> 3) * W_ROW_PITCH + (lt & 7) * 16;          // piece j: + j * 16 rows
        u32x4 ra[PW], rb[PW];                                               // even / odd stages in flight
        auto issue = [&](u32x4 (&rr)[PW], long g) __attribute__((always_inline)) {
            const u32x4* src = wg + (size_t)(g % NST_TILE) * W_PIECES + lt;
#pragma unroll
            for (int j = 0; j < PW; ++j) rr[j] = src[j * LOADER_THREADS];
        };
        auto commit = [&](const u32x4 (&rr)[PW], int buf) __attribute__((always_inline)) {
            unsigned char* dst = w_lds + buf * W_BUF_BYTES + dst0;
#pragma unroll
            for (int j = 0; j < PW; ++j) *(u32x4*)(dst + j * (LOADER_THREADS / 8) * W_ROW_PITCH) = rr[j];
        };
        // Stage g is multiplied out of buffer g&1.  While stage g runs, this wave puts stage g+2 in flight and writes
        // stage g+1 - fetched a whole stage earlier, so its L2 round trip is already over - into the other buffer,
        // whose last readers finished with the barrier that ended stage g-1.  Only weight loads live in this wave's
        // vmcnt queue, so the compiler's counted wait in front of the ds_writes never waits for HBM traffic.
        issue(ra, 0);
        commit(ra, 0);
        if (nstage_total > 1) issue(rb, 1);
        lds_done_then_barrier();                            // prologue barrier
        for (long g = 0; g < nstage_total; g += 2) {
            if (g + 2 < nstage_total) issue(ra, g + 2);
            if (g + 1 < nstage_total) commit(rb, 1);
            lds_done_then_barrier();                        // ends stage g
            if (g + 1 >= nstage_total) break;
            if (g + 3 < nstage_total) issue(rb, g + 3);
            if (g + 2 < nstage_total) commit(ra, 0);
            lds_done_then_barrier();                        // ends stage g+1
        }
        if constexpr (OFFLOAD) lds_done_then_barrier();     // final barrier (last tile staged)
        return;
    }

    // ==================================================================== COMPUTE (waves 0-3)
    const int r = lane & 31, hh = lane >> 5;
    const bool has_slope = p.slope != nullptr;
    const float slope = has_slope ? p.slope[0] : 0.f;
    const int a_off = r * W_ROW_PITCH + hh * 16;                                   // + buf, + (t*COUT + cb*32) rows, + ks*32
    const int b_off = ((2 * wave) * HALO_W + r) * PIX_PITCH + hh * 16;            // + buf, + tap, + pb row, + ks*32

    lds_done_then_barrier();                                // prologue barrier: stage 0 weights, phase 0 input, bias
    long gstage = 0, gphase = 0;
    for (int tl_i = 0; tl_i < ntl; ++tl_i) {
        f32x16 acc[NCB][2];
#pragma unroll
        for (int a = 0; a < NCB; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

        // OFFLOAD, COUT = 64: this tile's residual quads (the accumulator layout: 4 channels of one pixel per quad) are
        // fetched at the start of the tile's LAST phase, a whole phase (>= 3 stages) before the staging step adds them
        constexpr bool OFF_RES = OFFLOAD && COUT == 64;
        u32x2 resq[OFF_RES ? 16 : 1];
        float res_alpha = 1.f;

        constexpr int UNROLL_CHUNKS = OFFLOAD ? NCHUNK : 1;
#pragma unroll UNROLL_CHUNKS
        for (int chunk = 0; chunk < NCHUNK; ++chunk) {
            if constexpr (OFF_RES) {
                if (chunk == NCHUNK - 1 && p.res_mode != 0) {
                    const V3Tile c = make_tile(slot + (long)tl_i * G);
                    const unsigned char* rbase = (const unsigned char*)p.res + (size_t)c.m * hw * 64 * ES;     // res_mode 1
                    if (p.res_mode == 3) {
                        const int b = c.m / p.out_h, i = c.m - b * p.out_h;
                        if (p.alphas) res_alpha = p.alphas[(size_t)b * p.alpha_vs + (p.pair_last - i)];
                        rbase = (const unsigned char*)p.res + ((size_t)b * p.res_vs + i) * hw * 64 * ES;
                    }
                    const int gx = c.x0 + r, gxc = gx < W ? gx : W - 1;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int pb = q >> 3, co = ((q >> 2) & 1) * 32 + 8 * (q & 3) + 4 * hh;
                        const int gy = c.y0 + 2 * wave + pb, gyc = gy < H ? gy : H - 1;       // clamped: loads are unconditional
                        resq[q] = *(const u32x2*)(rbase + (((size_t)gyc * W + gxc) * 64 + co) * ES);
                    }
                }
            }
            const unsigned char* xin = in_lds + (int)(gphase & 1) * IN_BUF_BYTES + b_off;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const unsigned char* wst = w_lds + (int)(gstage & 1) * W_BUF_BYTES + a_off;
                // One compute wave per SIMD: nothing else hides the LDS round trip, so the fragment reads run DEPTH k-steps
                // ahead of the MFMAs that consume them, through a ring of DEPTH+1 fragment sets.  The sched_barriers pin
                // that order (left alone, hipcc re-uses one fragment register set and waits on every read it just issued).
                constexpr int NK = TPS * 4;                 // k-steps in this stage
                constexpr int DEPTH = COUT == 64 ? 2 : 1;   // COUT=128: 8 MFMA (256 cycles) per k-step cover one round trip
                bf16x8 fa[DEPTH + 1][NCB], fb[DEPTH + 1][2];
                auto load_k = [&](int i, int slot_) __attribute__((always_inline)) {
                    const int t = i >> 2, ks = i & 3;
                    const int tap = st * TPS + t;
                    const int ky = tap / 3, kx = tap - ky * 3;
                    const unsigned char* xb = xin + (ky * HALO_W + kx) * PIX_PITCH + ks * 32;
                    const unsigned char* wb = wst + t * COUT * W_ROW_PITCH + ks * 32;
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) fa[slot_][cb] = *(const bf16x8*)(wb + cb * 32 * W_ROW_PITCH);
                    fb[slot_][0] = *(const bf16x8*)(xb);
                    fb[slot_][1] = *(const bf16x8*)(xb + HALO_W * PIX_PITCH);
                };
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) load_k(i, i);
#pragma unroll
                for (int i = 0; i < NK; ++i) {
                    if (i + DEPTH < NK) load_k(i + DEPTH, (i + DEPTH) % (DEPTH + 1));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i % (DEPTH + 1)][cb], fb[i % (DEPTH + 1)][0], acc[cb][0], 0, 0, 0);
                        acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i % (DEPTH + 1)][cb], fb[i % (DEPTH + 1)][1], acc[cb][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                lds_done_then_barrier();
                ++gstage;
            }
            ++gphase;
        }

        if constexpr (OFFLOAD) {
            // ---- tile done: bias + PReLU (+ residual) -> one bf16 rounding -> staging tile (the input buffer this tile has
            // just released).  The INPUT waves store it after the next barrier while these waves multiply the next tile.
            unsigned char* stg0 = in_lds + (int)((gphase - 1) & 1) * IN_BUF_BYTES;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                unsigned char* rowp = stage_row(stg0, (2 * wave + pb) * 32 + r);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = cb * 32 + 8 * g + 4 * hh;
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[cb][pb][4 * g + j];
                        v += *(const f32x4*)(bias_lds + co);
                        if (has_slope) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] >= 0.f ? v[j] : slope * v[j];
                        }
                        if constexpr (OFF_RES) {
                            if (p.res_mode != 0) {
                                const u32x2 rq = resq[pb * 8 + cb * 4 + g];
                                f32x4 rr;
                                rr[0] = __uint_as_float(rq[0] << 16); rr[1] = __uint_as_float(rq[0] & 0xffff0000u);
                                rr[2] = __uint_as_float(rq[1] << 16); rr[3] = __uint_as_float(rq[1] & 0xffff0000u);
                                v = p.res_mode == 3 ? rr + res_alpha * v : v + rr;
                            }
                        }
                        u32x2 u;
                        u[0] = pack2_bf16(v[0], v[1]);
                        u[1] = pack2_bf16(v[2], v[3]);
                        *(u32x2*)(rowp + co * 2) = u;
                    }
                }
            }
        } else
        // ---- epilogue: bias, PReLU, residual, NHWC store
        {
            const V3Tile cur = make_tile(slot + (long)tl_i * G);
            const int m = cur.m;
            size_t out_img = (size_t)m;
            float alpha = 1.f;
            const unsigned char* res3 = nullptr;
            if (p.out_h > 0) {
                const int b = m / p.out_h, i = m - b * p.out_h;
                out_img = (size_t)b * p.out_vs + i;
                if (p.res_mode == 3) {
                    if (p.alphas) alpha = p.alphas[(size_t)b * p.alpha_vs + (p.pair_last - i)];
                    res3 = (const unsigned char*)p.res + ((size_t)b * p.res_vs + i) * hw * COUT * ES;
                }
            }
            unsigned char* outp = (unsigned char*)p.out + out_img * hw * COUT * ES;
            // The accumulator layout gives a lane 4 channels of ONE pixel, i.e. a wave store would touch 32 different
            // 128-byte lines with 16 bytes each.  So every (pixel block, 64-channel half) goes through a wave-private f32
            // staging tile in LDS - a quarter of the input buffer this tile has just released; the INPUT waves overwrite it
            // only at the end of the next phase - and comes back row-major: 16 lanes x 8 B (4 channels) = one full
            // 128-byte line per pixel, 4 lines per load / store instruction, residual read the same way.  f32 staging
            // keeps the single bf16 rounding of the direct store.
            constexpr int SROW = 64 * 4 + 16;                                   // staged row: 64 f32 + pad
            unsigned char* stg = in_lds + (int)((gphase - 1) & 1) * IN_BUF_BYTES + wave * (IN_BUF_BYTES / 4);
            static_assert(32 * SROW <= IN_BUF_BYTES / 4, "staging tile must fit a quarter of an input buffer");
            const int prow = lane >> 4, q4 = (lane & 15) * 4;                   // read-back: pixel-in-group, first channel
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int gy = cur.y0 + 2 * wave + pb;
                const int gyc = gy < H ? gy : H - 1;
#pragma unroll
                for (int ch = 0; ch < COUT / 64; ++ch) {
#pragma unroll
                    for (int cbl = 0; cbl < 2; ++cbl) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = cbl * 32 + 8 * g + 4 * hh;          // channel inside this half
                            f32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = acc[ch * 2 + cbl][pb][4 * g + j];
                            v += *(const f32x4*)(bias_lds + ch * 64 + col);
                            if (has_slope) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = v[j] >= 0.f ? v[j] : slope * v[j];
                            }
                            *(f32x4*)(stg + r * SROW + col * 4) = v;
                        }
                    }
                    // residual: all 8 row loads go out first, unconditionally (clamped addresses), so that their HBM round
                    // trips overlap instead of being serialised one per row behind a bounds branch
                    f32x4 resv[8];
                    if (p.res_mode != 0) {
                        const unsigned char* rbase = p.res_mode == 1 ? (const unsigned char*)p.res + (size_t)m * hw * COUT * ES
                                                   : p.res_mode == 2 ? (ch == 0 ? cur.src0 : cur.src1) : res3;
                        const int rpitch = p.res_mode == 2 ? 64 : COUT;
                        const int rch = p.res_mode == 2 ? 0 : ch * 64;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int gx = cur.x0 + 4 * i + prow;
                            const int gxc = gx < W ? gx : W - 1;
                            resv[i] = load4<HRN_BF16>(rbase, ((size_t)gyc * W + gxc) * rpitch + rch + q4);
                        }
                    }
                    // same wave wrote and reads: LDS executes a wave's instructions in order, no barrier needed
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int px = 4 * i + prow;
                        f32x4 v = *(const f32x4*)(stg + px * SROW + q4 * 4);
                        const int gx = cur.x0 + px;
                        if (p.res_mode == 3) v = resv[i] + alpha * v;
                        else if (p.res_mode != 0) v += resv[i];
                        if (gy < H && gx < W) store4<HRN_BF16>(outp, ((size_t)gy * W + gx) * COUT + ch * 64 + q4, v);
                    }
                }
            }
        }
    }
    if constexpr (OFFLOAD) lds_done_then_barrier();         // final barrier: hands the last staging tile to the INPUT waves
}


template <int CIN, int COUT, bool OFFLOAD>
int launch_v3(const ConvParams& p, hipStream_t stream) {
    constexpr int LDS_BYTES = V3Geom<COUT, OFFLOAD>::LDS_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    { const int rc_lds = hrn_allow_lds((const void*)conv3x3_v3_kernel<CIN, COUT, OFFLOAD>, LDS_BYTES); if (rc_lds) return rc_lds; }
    const long tiles = (long)((p.W + CONV_TILE_W - 1) / CONV_TILE_W) * ((p.H + CONV_TILE_H - 1) / CONV_TILE_H);
    const long total = tiles * p.M;
    HRN_CHECK(total > 0 && total < (1L << 40), -2, "conv3x3_v3: bad tile count %ld", total);
    long grid = hrn_device_cus();                   // one persistent 8-wave workgroup per CU
    if (total < grid) grid = total;
    if (grid >= 8) grid &= ~7L;
    static const char* fam = CIN == 64 ? "conv3x3_bf16_64x64" : (COUT == 64 ? "conv3x3_bf16_128x64" : "conv3x3_bf16_128x128");
    static const char* fam_res = CIN == 64 ? "conv3x3_bf16_64x64+res" : (COUT == 64 ? "conv3x3_bf16_128x64+res" : "conv3x3_bf16_128x128+res");
    const double px = (double)p.M * p.H * p.W;
    HrnProfScope prof(p.res_mode ? fam_res : fam, 2.0 * CIN * COUT * 9 * px, px * 2 * (CIN + COUT + (p.res_mode ? COUT : 0)), stream);
    hipLaunchKernelGGL((conv3x3_v3_kernel<CIN, COUT, OFFLOAD>), dim3((unsigned)grid), dim3(512), LDS_BYTES, stream, p);
    HRN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// bf16, no folded scale / ReLU (HRNet layers).  Returns -100 when the shape is not covered (caller falls back).
int hrn_launch_conv3x3_v3(int cin, int cout, const ConvParams& p, hipStream_t stream) {
    if (p.scale || p.relu) return -100;
    // HRN_CONV_OFFLOAD=0 keeps the whole epilogue on the compute waves (A/B timing); default: off-loaded stores wherever
    // the kernel supports them (every COUT = 64 layer; COUT = 128 layers without a residual)
    static int offload = -1;
    if (offload < 0) { const char* e = getenv("HRN_CONV_OFFLOAD"); offload = e ? atoi(e) : 1; }
    if (cin == 64 && cout == 64) {
        // resident-weights kernel (conv3x3_r64.hip) for the encoder's plain 64 -> 64 layers; HRN_CONV_R64=0 for A/B timing
        static int r64 = -1;
        if (r64 < 0) { const char* e = getenv("HRN_CONV_R64"); r64 = e ? atoi(e) : 1; }
        if (r64) { const int rc = hrn_launch_conv3x3_r64(p, stream); if (rc != -100) return rc; }
        return offload ? launch_v3<64, 64, true>(p, stream) : launch_v3<64, 64, false>(p, stream);
    }
    if (cin == 128) {
        // conv3x3_v6.hip for the three layers of a fusion level; HRN_CONV_V6=0 runs this file's kernel instead (A/B timing)
        static int v6 = -1;
        if (v6 < 0) { const char* e = getenv("HRN_CONV_V6"); v6 = e ? atoi(e) : 1; }
        if (v6) { const int rc = hrn_launch_conv3x3_v6(cout, p, stream); if (rc != -100) return rc; }
    }
    // what conv3x3_r64 / v6 decline (images beyond their 32-bit in-image offsets, > 8.3 Mpixel) runs on this file's kernel
    if (cin == 128 && cout == 64) return offload ? launch_v3<128, 64, true>(p, stream) : launch_v3<128, 64, false>(p, stream);
    if (cin == 128 && cout == 128)
        return (offload && p.res_mode == 0) ? launch_v3<128, 128, true>(p, stream) : launch_v3<128, 128, false>(p, stream);
    return -100;
}
