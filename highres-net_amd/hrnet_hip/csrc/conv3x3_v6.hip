// conv3x3 128 -> {128, 64}, bf16: the three layers of a fusion level (HRNet.py:90-97, :113-131) on v_mfma_f32_16x16x32_bf16
// (under load the chip holds a higher clock on this shape than on 32x32x16: CDNA4 guide, "DVFS give-back" item 7; the same skeleton
// on 32x32x16 was A/B-timed and removed).  512-pixel tiles (16 x 32), 8 MFMA waves, two per SIMD, LDS-DMA
// staging, one barrier per stage.
//   operands   the PIXELS are the MFMA's A operand (16 pixels x 32 cin: lane l reads pixel l & 15, 16-byte chunk q = l >> 4 of its
//              64-byte row), the weights its B operand (16 couts x 32 cin, same shape).  K = 32 is exactly one staged chunk: one
//              k-step per tap.
//   swizzle    physical chunk = q ^ (((row >> 2) & 1) << 1) for both LDS images (conflict-free ds_read_b128 for every base
//              alignment; found by enumeration), applied on the DMA source side and on the read.
//   per wave   64 pixels (rows 2w, 2w+1; 4 blocks of 16) x COUT couts (NCB = COUT / 16 blocks of 16) = COUT accumulator registers;
//              a step = one tap x one pair of cout blocks: 2 weight fragment reads (+ 4 pixel fragments once per tap,
//              double-buffered) for 8 MFMAs.
//   couts      row r of cout block cb holds cout NCB * r + cb (the interleave is applied by the weight DMA's lane offsets; the
//              packed weights are unchanged).  With the pixels on the A side, acc[cb][pxb][e] of lane (q = l >> 4, c15 = l & 15) is
//              pixel 4q + e of pixel block pxb, channel NCB * c15 + cb: the NCB values a lane holds of one pixel are contiguous
//              bytes of that pixel's row, the 16 lanes of a q together are the whole row.
// What the kernel does about the costs its predecessors' stamps showed (round 1: conv3x3_v4 / v5, profiles/
// r01_final_inkernel_stamps.txt; round 2's first form of this file, profiles/r02_final_v6_stamps.txt):
//   * halo DMA through a buffer descriptor (buffer_load_dwordx4 ... lds): the per-lane byte offsets of a wave's <= 5 pieces
//     are computed ONCE per tile (5 registers); a lane whose halo pixel lies outside the image carries offset 0x80000000,
//     which the descriptor's range check turns into zeros written to LDS (checked on the hardware: scratch micro-test,
//     DESIGN.md section 3.1).  No address arithmetic, no bounds tests and no zero page per piece; the channel chunk and the
//     weight stage travel in the scalar offset.  Weights go through a descriptor too (lane offset constant).
//   * every DMA is issued from an MFMA gap of the stage (one piece after every eighth MFMA), by all eight waves alike.
//   * the epilogue works straight from the accumulators: activation, residual, bf16 packing and the store need no LDS staging
//     and no lane exchange, and one store (or residual load) instruction covers four whole pixel rows.  Stores go through a
//     buffer descriptor (a pixel outside the image gets an offset the range check drops: no branches).  Round 0 of the
//     residual (the pair gather z, or s_i of the view stack) is fetched by LDS-DMA into 4 KB per wave while the tile's last
//     stage computes (no registers for it beside the fragments) - it comes from HBM, several thousand cycles away under this
//     load -, rounds 1-2 go into the dead fragment registers when the epilogue starts, round 3 follows round 0.
//   * two weight ring slots: stage s+1's weights are issued first thing in stage s and waited for at its end (L2 hits).
// RESM: 0 none | 2 the pair gather z (COUT = 128: t2 = z + PReLU(conv(t1))) | 3 s_i + alpha_partner * f into the view stack
// (COUT = 64, HRNet.py:123-131).
// LDS (COUT = 128): 2 x 24,576 (weights) + 2 x 39,936 (halo) + 512 (bias) + 8 x 4,096 (residual round 0) = 162,304 B.
// Ordering rules (guide, "Pipelining across barriers"): a wave waits for its own DMAs with a counted vmcnt BEFORE the
// barrier that precedes the stage reading them; a buffer is re-filled only after a barrier every reader of its previous
// contents has passed.  vmcnt counts in issue order, so a stage issues its weights first and its halo pieces last: waiting
// until only the halo pieces are outstanding retires the weights and everything older.
#include "conv3x3_v6_impl.h"

#ifdef V6_STAMP
__device__ unsigned long long hrn_v6_stamps[256 * 8 * 24];
extern "C" int hrn_dbg_read_stamps_v6(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hrn_v6_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

// bf16, 128 input channels.  COUT = 128: residual none or the pair gather (res_mode 2); COUT = 64: none or the alpha residual into
// the view stack (res_mode 3).  Returns -100 when not applicable.
int hrn_launch_conv3x3_v6(int cout, const ConvParams& p, hipStream_t stream) {
    if (p.scale || p.relu) return -100;
    if (cout != 64 && cout != 128) return -100;
    if (cout == 128 && p.res_mode != 0 && p.res_mode != 2) return -100;
    if (cout == 64 && ((p.res_mode != 0 && p.res_mode != 3) || p.in_pair)) return -100;
    if ((p.in_pair || p.res_mode == 2) && p.pair_h <= 0) return -100;
    if (p.res_mode == 3 && (p.out_h <= 0 || !p.res)) return -100;
    long grid = 0;
    { const int rc = v6_grid(p, 128, grid); if (rc) return rc; }
    const double px = (double)p.M * p.H * p.W;
    const char* fam = cout == 128 ? (p.res_mode ? "conv3x3_bf16_128x128+res" : "conv3x3_bf16_128x128")
                                  : (p.res_mode ? "conv3x3_bf16_128x64+res" : "conv3x3_bf16_128x64");
    HrnProfScope prof(fam, 2.0 * 128 * cout * 9 * px, px * 2 * (128 + cout + (p.res_mode ? cout : 0)), stream);
    if (cout == 128) {
        if (p.in_pair) return p.res_mode ? launch_v6<128, 128, 2, true, false>(p, grid, stream) : launch_v6<128, 128, 0, true, false>(p, grid, stream);
        return p.res_mode ? launch_v6<128, 128, 2, false, false>(p, grid, stream) : launch_v6<128, 128, 0, false, false>(p, grid, stream);
    }
    return p.res_mode ? launch_v6<128, 64, 3, false, false>(p, grid, stream) : launch_v6<128, 64, 0, false, false>(p, grid, stream);
}
