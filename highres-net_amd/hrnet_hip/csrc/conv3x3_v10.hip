// conv3x3_v10: the bf16 128 -> {128, 64} convolutions of a fusion level (HRNet.py:17-22,93-97,114-131) with the two wave groups of a
// workgroup ONE STAGE APART, so that a group's epilogue runs beside the other group's MFMAs instead of beside its epilogue.
//
// conv3x3_v6 (same tile, same per-wave work: DESIGN.md 3.1) spends 19 % of a launch in the epilogue: all eight waves leave the main
// loop together, move 128 KB of residual in and 128 KB of output out through the CU's one vector-memory path, and the matrix pipe
// idles meanwhile (profiles/r03_v6_packed_epilogue_ab.txt).  Hiding that needs another tile's MFMAs on the same SIMD.  Two teams
// on two tiles (conv3x3_v9, round 3) doubled the weight stream and lost.  Here both groups stay on ONE 16 x 32 tile and ONE weight
// stream; only the clock of group B (waves 4-7, tile rows 8-15) is set back by one stage against group A (waves 0-3, rows 0-7):
//
//   step tau (one workgroup barrier per step):   A computes stage tau,   B computes stage tau - 1      (12 stages per tile)
//   a group's epilogue of tile t is the first thing it does in the step of its stage 0 of tile t + 1 - A's in step 12 (t + 1), B's in
//   step 12 (t + 1) + 1 - while the other group is in its last / second stage: the epilogues of the two groups never meet, and each
//   has a whole stage of the partner's MFMAs beside it.
//
// What that costs in LDS: a weight stage is read by A in step s and by B in step s + 1, so the ring has THREE slots (stage s + 1 is
// fetched during step s into the slot B left at the end of step s - 1); a halo chunk k is read by A in steps 3k .. 3k + 2 and by B in
// 3k + 1 .. 3k + 3, so chunk k + 2 (same buffer) can only be fetched from step 3k + 4 on: all of a chunk's pieces are issued in the
// middle stage of the previous chunk (v6: first and middle) and waited for at the end of its last stage.  3 x 24 KB + 2 x 39 KB + bias =
// 150.5 KB; v6's 32 KB residual FIFO has no room and no purpose any more (round 0 of the residual comes from HBM like the others:
// its latency is now beside the partner's MFMAs).
// The tap row a wave works on is a run-time value here (A and B differ), so the halo swizzle depends on the pixel's COLUMN only
// (v6: on the pixel index): a tap row is then a pure byte offset, added to six fragment addresses per stage, and one stage body serves
// every tap row (v6 unrolls three).
#include <type_traits>
#include "conv3x3.h"

#ifndef V10_ABL
#define V10_ABL 0          // timing-only ablation (results are WRONG when set): 2 = no epilogue
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
constexpr int NSTAGE = 3 * NCH;                             // stages per tile

template <int COUT> struct G10 {
    static constexpr int NCB = COUT / 16;                   // cout blocks of 16 per wave
    static constexpr int NQ = NCB / 2;                      // steps per tap (2 cout blocks x 4 pixel blocks = 8 MFMAs each)
    static constexpr int TAP_BYTES = COUT * 64;             // one tap x 32 cin
    static constexpr int WST = 3 * TAP_BYTES;               // one stage: 24,576 | 12,288
    static constexpr int W_PIECES = WST / 1024;             // 24 | 12
    static constexpr int OFF_IN = 3 * WST;                  // behind the three-slot weight ring
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
__global__ __launch_bounds__(512, 2) void conv3x3_v10_kernel(const ConvParams p) {
    typedef G10<COUT> GEO;
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
    const int grp = w >> 2;                                  // 0: group A, 1: group B (one stage behind)
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
    constexpr int N_ITEMS = NW3 + 5;                        // DMA items of a step: the next stage's weights, then (middle stage of a chunk) the next halo chunk
    static_assert(N_ITEMS <= NSTEP + 2, "one DMA item per MFMA step, the last step takes what is left");
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

    // ---- DMA side: the tile whose halo chunks are being fetched
    unsigned dma_m = slot0 / tiles, dma_t = slot0 - dma_m * tiles;
    size_t inA, inB;
    in_bases(dma_m, inA, inB);
    tile_offsets(dma_t);
    // ---- compute side: the tile this wave's group is working on, and what its epilogue needs
    unsigned own_m = dma_m, own_t = dma_t;
    int y0 = 0, x0 = 0;
    const unsigned char *resA = nullptr, *resB = nullptr;
    unsigned char* outp = nullptr;
    float res_alpha = 1.f;
    auto own_geometry = [&]() __attribute__((always_inline)) {
        const int ty_ = own_t / tiles_x;
        y0 = ty_ * TH; x0 = (own_t - ty_ * tiles_x) * TW;
        size_t oimg = own_m;
        res_alpha = 1.f;
        if (p.out_h > 0) {
            const unsigned ob = own_m / (unsigned)p.out_h, oi = own_m - ob * (unsigned)p.out_h;
            oimg = (size_t)ob * p.out_vs + oi;
            if (RESM == 3) {
                resA = resB = (const unsigned char*)p.res + ((size_t)ob * p.res_vs + oi) * hw * 128;
                if (p.alphas) res_alpha = p.alphas[(size_t)ob * p.alpha_vs + (p.pair_last - oi)];
            }
        }
        if (RESM == 2) {
            const unsigned bb = own_m / (unsigned)p.pair_h, i = own_m - bb * (unsigned)p.pair_h;
            resA = (const unsigned char*)p.stack + ((size_t)bb * p.pair_vs + i) * hw * 128;
            resB = (const unsigned char*)p.stack + ((size_t)bb * p.pair_vs + (p.pair_last - i)) * hw * 128;
        }
        outp = (unsigned char*)p.out + oimg * hw * ROW;
    };

    // ---- the epilogue of the group's tile: registers and global memory only.  acc[cb][pxb][e] of lane (q, c15) is pixel 4q + e of
    // pixel block pxb, channel NCB c15 + cb: the NCB values a lane holds of one pixel are LB contiguous bytes of its row, sixteen lanes
    // the whole row (conv3x3_v6.hip).  The residual of round r (= pixel block r), piece j: lane (q, c15) fetches its own share of pixel
    // 4q + j - of z = cat(view i, partner) (64 channels = 128 bytes each) the 16 bytes that hold channels 8 c15 .. 8 c15 + 7.
    auto epilogue = [&]() __attribute__((always_inline)) {
        // nothing of this wave is in flight here (the previous step ended on vmcnt(0)), but hipcc cannot see inline-asm waits: a wait it
        // CAN see lets it count the residual rounds instead of answering their first use with vmcnt(0)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        int le = lane;
        asm volatile("" : "+v"(le));                        // every lane-derived address below is formed here, per tile
        const int c15e = le & 15, qe = le >> 4;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, (int)(hw * ROW), 0x00020000);
        auto res_src = [&](int r, int j) __attribute__((always_inline)) -> const unsigned char* {
            const int gy = y0 + 2 * w + (r >> 1), gyc = gy < H ? gy : H - 1;
            const int gx = x0 + 16 * (r & 1) + 4 * qe + j, gxc = gx < W ? gx : W - 1;
            const unsigned char* view = (RESM == 2 && c15e >= 8) ? resB : resA;
            return view + ((unsigned)((gyc * W + gxc) * 128) + (RESM == 2 ? (unsigned)((c15e & 7) * 16) : (unsigned)(c15e * LB)));
        };
        lane_row_t rq[4][4];                                // [round][j]
        auto res_load = [&](int r) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rq[r][j] = __builtin_nontemporal_load((const lane_row_t*)res_src(r, j));
        };
#ifndef V10_RES_AHEAD
#define V10_RES_AHEAD 3                                     // residual rounds in flight when the epilogue starts (v6: 2, in the dead fragment registers)
#endif
#ifndef V10_EPI_PRIO
#define V10_EPI_PRIO 3
#endif
        if (V10_EPI_PRIO) __builtin_amdgcn_s_setprio(V10_EPI_PRIO);     // the partner has a stage's worth of slack in this step, this wave does not
        if (RES) {
#pragma unroll
            for (int r = 0; r < V10_RES_AHEAD; ++r) res_load(r);
        }
        __builtin_amdgcn_sched_barrier(0);                  // (hipcc would otherwise start all four rounds' loads here and spill them)
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
                const unsigned voff = (unsigned)((gy * W + gx) * ROW + c15e * LB) | ((unsigned)(W - 1 - gx) & OOB) | (gy < H ? 0u : OOB);
                if constexpr (LB == 16) __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, voff, 0, 2);      // nt: the next launch reads it from HBM anyway
                else __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, voff, 0, 2);
            }
            if (RES && r + V10_RES_AHEAD < 4) res_load(r + V10_RES_AHEAD);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (V10_EPI_PRIO) __builtin_amdgcn_s_setprio(0);
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

    const int NT = NSTAGE * ntl;                             // stages of this workgroup
    const int n_chunks = NCH * ntl;
    // A's position = the step: tau = 3 kA + tgA, chunk ccA = kA % 4 of its tile, weight slot slotA = tau % 3
    int tgA = 0, ccA = 0, kA = 0, slotA = 0;
    // this wave's own position (its stage sig = tau - grp): tap row, chunk, weight slot, halo buffer
    int tgw = 0, ccw = 0, slotw = 0, bufw = 0;

    // Steps 0 .. NT.  Every wave runs the stage body in every step - group B's in step 0 (before its stage 0) and group A's in step NT
    // (behind its last stage) work on whatever the LDS holds and their sums are thrown away (B's accumulators are initialised in step 1,
    // A's last epilogue has run by then): two idle stages per workgroup LIFETIME are cheaper than a second instance of the body, and
    // an accumulator array that lives in one set of registers on every path is what keeps hipcc from copying and spilling it.
    for (int tau = 0; tau <= NT; ++tau) {
        const int sig = tau - grp;
        const bool active = sig >= 0 && sig < NT;
        if (sig >= 0 && sig <= NT && tgw == 0 && ccw == 0) {            // sig % 12 == 0: between two tiles of this group
            if (sig > 0) {
                if (!(V10_ABL & 2)) epilogue();
                else {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int pxb = 0; pxb < 4; ++pxb) asm volatile("" :: "v"(acc[cb][pxb]));
                }
                next_tile(own_m, own_t);
            }
            if (sig < NT) {
                own_geometry();
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
        }
        // ---- DMA duties of the step, the same for every wave: the weights of stage tau + 1, and in the middle stage of A's chunk the
        // whole of halo chunk kA + 1 (its buffer was B's until the end of the previous step)
        const bool w_next = tau + 1 < NT;
        const bool halo_step = tgA == 1 && kA + 1 < n_chunks;
        if (halo_step && ccA == NCH - 1) {                  // that chunk is chunk 0 of the next tile
            next_tile(dma_m, dma_t);
            in_bases(dma_m, inA, inB);
            tile_offsets(dma_t);
        }
        const int tg2 = tgA == 2 ? 0 : tgA + 1;
        const int c2 = tgA == 2 ? ((ccA + 1) & (NCH - 1)) : ccA;
        const int slot2 = slotA == 2 ? 0 : slotA + 1;
        const int cn = (ccA + 1) & (NCH - 1), nbuf = (kA + 1) & 1;
        const size_t hb = (PAIR && cn >= 2) ? inB : inA;
        const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)(src0 + hb), 0, (int)img_bytes, 0x00020000);
        auto issue_item = [&](auto it_c) __attribute__((always_inline)) {
            constexpr int it = decltype(it_c)::value;
            if constexpr (it < NW3) { if (w_next) dma_w(c2, tg2, slot2, it); }
            else if constexpr (it < N_ITEMS) { if (halo_step) dma_halo(rs_h, cn, nbuf, it - NW3, hoff[it - NW3]); }
        };

        {
            // ---- NSTEP steps = 3 taps x NQ cout pairs, 8 MFMAs each ((k, pxb): cout block 2qt+k x pixel block pxb).  Hand-issued fragment
            // reads with counted waits (conv3x3_v6.hip): prologue B0..B3(tap 0), A0(0), A1(0); step i: A0(i+1) after MFMA 0, A1(i+1) after
            // MFMA 1, and in the second step of a tap the next tap's B0..B3 after MFMAs 2..5; one DMA item behind the last MFMA of a step.
            const unsigned abase = a_off + (unsigned)(slotw * WST);
            const unsigned boff = (unsigned)(bufw * IN_BYTES + tgw * ROWB);
            unsigned bcur[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) bcur[j][kx] = baddr0[j][kx] + boff;
            bf16x8 fa[2][2], fb[2][4];
            auto rd = [&](bf16x8& dst, unsigned addr, int imm) __attribute__((always_inline)) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm));
            };
            auto load_b1 = [&](int tap, int pxb) __attribute__((always_inline)) {
                rd(fb[tap & 1][pxb], bcur[pxb >> 1][tap], (pxb & 1) * 1024);
            };
            auto load_a1 = [&](int i, int k) __attribute__((always_inline)) {       // step i = (tap i / NQ, cout pair i % NQ)
                rd(fa[i & 1][k], abase, (i / NQ) * TAP_BYTES + (i % NQ) * 2048 + k * 1024);
            };
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
                    if (g == 7) {                           // (items in issue order = age order: the counted wait at the step's end relies on it)
                        issue_item(std::integral_constant<int, i>{});
                        if constexpr (i == NSTEP - 1) static_for<N_ITEMS - NSTEP>([&](auto e_c) __attribute__((always_inline)) {
                            issue_item(std::integral_constant<int, NSTEP + decltype(e_c)::value>{});
                        });
                    }
                    if (BAL && g == 7 && i == (NSTEP * BAL) / 8 - 1 && w >= 4) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        // the next stage's weights (and every older access, the epilogue's stores included) have landed once only this step's halo pieces
        // are outstanding; the halo chunk itself is waited for at the end of the chunk's last stage
        wait_vm_rt(halo_step ? n_in : 0);
        wg_barrier();
        if (active) {
            if (++tgw == 3) { tgw = 0; ccw = (ccw + 1) & (NCH - 1); bufw ^= 1; }
            if (++slotw == 3) slotw = 0;
        }
        if (++tgA == 3) { tgA = 0; ccA = (ccA + 1) & (NCH - 1); ++kA; }
        if (++slotA == 3) slotA = 0;
    }
    if (grp && !(V10_ABL & 2)) epilogue();                  // group B's last tile (its stage NT - 1 ran in step NT)
    wait_vm<0>();                                           // nothing of this workgroup may still be in flight when it ends
}

template <int COUT, int RESM, bool PAIR>
int launch_v10(const ConvParams& p, long grid, hipStream_t stream) {
    typedef G10<COUT> GEO;
    static_assert(GEO::LDS_BYTES <= 160 * 1024, "LDS budget");
    { const int rc_lds = hrn_allow_lds((const void*)conv3x3_v10_kernel<COUT, RESM, PAIR>, GEO::LDS_BYTES); if (rc_lds) return rc_lds; }
    hipLaunchKernelGGL((conv3x3_v10_kernel<COUT, RESM, PAIR>), dim3((unsigned)grid), dim3(512), GEO::LDS_BYTES, stream, p);
    HRN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int hrn_launch_conv3x3_v10(int cout, const ConvParams& p, hipStream_t stream) {
    if (p.scale || p.relu) return -100;
    if (cout != 64 && cout != 128) return -100;
    if (cout == 128 && p.res_mode != 0 && p.res_mode != 2) return -100;
    if (cout == 64 && ((p.res_mode != 0 && p.res_mode != 3) || p.in_pair)) return -100;
    if ((p.in_pair || p.res_mode == 2) && p.pair_h <= 0) return -100;
    if (p.res_mode == 3 && (p.out_h <= 0 || !p.res)) return -100;
    const long tiles = (long)((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH);
    const long total = tiles * p.M;
    HRN_CHECK(total > 0, -2, "conv3x3_v10: bad tile count %ld", total);
    if (total >= (1L << 30) || (long)p.H * p.W * 256 >= (1L << 31)) return -100;     // 32-bit tile / in-image byte arithmetic
    long grid = hrn_device_cus();
    if (total < grid) grid = total;
    if (grid >= 8) grid &= ~7L;
    const double px = (double)p.M * p.H * p.W;
    const char* fam = cout == 128 ? (p.res_mode ? "conv3x3_bf16_128x128+res" : "conv3x3_bf16_128x128")
                                  : (p.res_mode ? "conv3x3_bf16_128x64+res" : "conv3x3_bf16_128x64");
    HrnProfScope prof(fam, 2.0 * 128 * cout * 9 * px, px * 2 * (128 + cout + (p.res_mode ? cout : 0)), stream);
    if (cout == 128) {
        if (p.in_pair) return p.res_mode ? launch_v10<128, 2, true>(p, grid, stream) : launch_v10<128, 0, true>(p, grid, stream);
        return p.res_mode ? launch_v10<128, 2, false>(p, grid, stream) : launch_v10<128, 0, false>(p, grid, stream);
    }
    return p.res_mode ? launch_v10<64, 3, false>(p, grid, stream) : launch_v10<64, 0, false>(p, grid, stream);
}
