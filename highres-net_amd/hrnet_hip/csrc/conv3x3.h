// 3x3 / pad 1 / stride 1 convolution as an LDS-staged implicit GEMM on MFMA (gfx950).
#pragma once
#include "common.h"

// Tile geometry shared by the kernel, the weight packer and the host launcher.
#define CONV_TILE_H 8
#define CONV_TILE_W 32
#define CONV_STEP_BYTES 8192   // one weight slice: 64 output channels x 128 bytes of K

struct ConvParams {
    const void* in;        // in_pair == 0: [M][H][W][CIN];  in_pair == 1: unused (input gathered from `stack`)
    const void* stack;     // view stack [B][pair_vs][H][W][64] for the pair descriptor (pair_h > 0)
    int in_pair;           // 1: the conv input is cat(view i, view pair_last - i) on channels, never materialised
    void* out;             // plain: [M][H][W][COUT]; slot: view stack, image b*out_vs + i
    const void* res;       // res_mode 1: [M][H][W][COUT]; res_mode 3: view stack, image b*res_vs + i
    const float* alphas;   // res_mode 3: [B][alpha_vs] (null -> scale 1)
    const void* wpk;       // packed weights, see conv_pack_index()
    const float* bias;     // [COUT] f32
    const float* slope;    // PReLU slope (1 float, device) or null = no activation
    int M, H, W;
    int pair_h, pair_last, pair_vs;   // pair_h > 0: image m -> (b = m / pair_h, i = m % pair_h); channels 0..63 = view i, 64..127 = view pair_last - i
    int out_h, out_vs;                // out_h > 0: output image m -> slot b*out_vs + i
    int res_mode;                     // 0 none | 1 tensor `res` | 2 pair gather of `in` | 3 x_i (slot of `res`) + alpha_partner * y
    int alpha_vs, res_vs;
    int relu;                         // 1: y = max(y, 0) after bias (ShiftNet eval with folded BN uses scale/shift below)
    const float* scale;               // optional per-channel scale applied before bias (folded BatchNorm), null = 1
    // HRN_BF16X3 only: every activation tensor is a PAIR of bf16 planes (hi, lo) in the bf16 layout; byte offset of the lo plane from
    // the hi plane of `in` / `stack` / `out` / `res` (0 otherwise)
    size_t in_lo, stack_lo, out_lo, res_lo;
    const float* only_if_nonpos;      // f32 kernel only: when set, the launch does nothing unless only_if_nonpos[0] <= 0 (the backward's
                                      // recomputation of a pre-activation, needed only behind a PReLU whose slope is not positive)
};

// Number of packed weight elements of one conv layer (== cin*cout*9).
static inline size_t conv_packed_elems(int cin, int cout) { return (size_t)cin * cout * 9; }

// Launch on `stream`.  dt: HRN_F32 / HRN_BF16; (cin, cout) in {64,128}^2.  Returns 0 or a negative error.
int hrn_launch_conv3x3(int dt, int cin, int cout, const ConvParams& p, hipStream_t stream);

// bf16 64 -> 64 with LDS-resident weights (conv3x3_r64.hip); -100 = not applicable, caller picks another kernel.
int hrn_launch_conv3x3_r64(const ConvParams& p, hipStream_t stream);

// bf16 128 -> {128, 64}: the three layers of a fusion level; descriptor-based halo DMA issued from the MFMA gaps, epilogue straight
// from the accumulators in whole pixel rows (conv3x3_v6.hip); -100 = not applicable.
int hrn_launch_conv3x3_v6(int cout, const ConvParams& p, hipStream_t stream);

// bf16x3 (split-bf16: hi/lo planes, three MFMAs per product) on the conv3x3_v6 skeleton: (cin, cout) = (64, 64) with res_mode 0 | 1,
// (128, 128) with 0 | 2 (+ pair-gather input), (128, 64) with 0 | 3 (conv3x3_v6x3.hip).  No other kernel implements this dtype.
int hrn_launch_conv3x3_v6x3(int cin, int cout, const ConvParams& p, hipStream_t stream);

// Pack OIHW f32 weights [cout][cin][3][3] into the kernel's step-major layout (device to device).
int hrn_launch_conv_pack(int dt, int cin, int cout, const float* w_oihw, void* packed, hipStream_t stream);
