// Launchers of the backward building blocks (backward.hip).  dt = HRN_F32 (default) or HRN_BF16X3: in the bf16x3 training mode every
// activation / gradient tensor is a pair of bf16 planes (hi, then lo directly behind it: a tensor of n elements has its lo plane 2 n
// bytes further on) behind the same `float*` arguments; parameters and their gradients are always f32.  Same conventions as kernels.h: asynchronous on `stream`, no
// allocation, no synchronisation; gradients are ACCUMULATED (+=) into their destination, like autograd's .grad.
#pragma once
#include "common.h"

// bytes of the `scratch` buffer the launchers below share (wgrad partial slabs, reduction partials)
size_t hrn_bwd_scratch_bytes(int num_cus);

// PReLU backward and the bias gradient of the convolution in front of it, in one pass: g = dy * PReLU'(x) ([rows][C], C in
// {64, 128}; may alias dy), dslope[0] += sum dy * min(x, 0), db[c] += sum_rows g[row][c].  x comes from the stored post-activation
// y when slope[0] > 0 and from the pre-activation xpre otherwise (decided on the device; the caller recomputes xpre with a launch
// gated the same way: ConvParams::only_if_nonpos)
int hrn_launch_prelu_bwd_bias(const float* dy, const float* y, const float* xpre, const float* slope, float* g, size_t rows, int C,
                              float* dslope, float* db, void* scratch, hipStream_t s, int dt = HRN_F32);
// db[c] += sum_rows g[row][c], C in {64, 128}
int hrn_launch_colsum(const float* g, size_t rows, int C, float* db, void* scratch, hipStream_t s, int dt = HRN_F32);
// wt[ci][co][ky][kx] = w[co][ci][2-ky][2-kx]: the OIHW tensor whose forward convolution is the data gradient
int hrn_launch_dgrad_weights(const float* w, float* wt, int cin, int cout, hipStream_t s);
// dw[co][ci][3][3] += sum g (x) shifted x; x plain [M][H][W][cin] or the pair gather of `stack` (cin = 128)
int hrn_launch_conv_wgrad(const float* x, const float* stack, int in_pair, int pair_h, int pair_last, int pair_vs, const float* g,
                          int M, int H, int W, int cin, int cout, float* dw, void* scratch, int num_cus, hipStream_t s);
// the same in the bf16x3 training mode (wgrad_x3.hip): x / stack and g are pairs of bf16 planes, x_lo / g_lo the byte offsets of their lo planes
int hrn_launch_conv_wgrad_x3(const void* x, const void* stack, size_t x_lo, int in_pair, int pair_h, int pair_last, int pair_vs, const void* g,
                             size_t g_lo, int M, int H, int W, int cin, int cout, float* dw, void* scratch, int num_cus, hipStream_t s);
// dW[co][ci][tap] += sum over workgroups of the partial slabs [nblk][9][64][64] of one (cout chunk, cin chunk) pair, fixed order
int hrn_launch_wgrad_finish(const float* partial, int nblk, float* dw, int cin, int co_chunk, int ci_chunk, hipStream_t s);
// stem 2 -> 64: in0 = image m (stride0 floats apart), in1 = plane m / rep1; dw [64][2][3][3]
int hrn_launch_stem_wgrad(const float* in0, size_t stride0, const float* in1, int rep1, size_t stride1, const float* g, int M, int H,
                          int W, float* dw, void* scratch, int num_cus, hipStream_t s, int dt = HRN_F32);
// the same with `sub` [M][2] subtracted from the in-image pixels of the two planes first (ShiftNet's mean-free input)
int hrn_launch_stem_wgrad_sub(const float* in0, size_t stride0, const float* in1, int rep1, size_t stride1, const float* sub,
                              const float* g, int M, int H, int W, float* dw, void* scratch, int num_cus, hipStream_t s, int dt = HRN_F32);
int hrn_launch_add(const float* a, const float* b, float* o, size_t n, hipStream_t s, int dt = HRN_F32);
// fusion level helpers (HRNet.py:113-132): forward update of the kept views, and the two backward maps
int hrn_launch_fuse_update(const float* stack, int n_in, const float* f, const float* alphas, int alpha_vs, int pair_last, int half,
                           int alpha_residual, float* out, size_t hw, int B, hipStream_t s, int dt = HRN_F32);
int hrn_launch_fuse_df(const float* dsn, const float* alphas, int alpha_vs, int pair_last, int half, int alpha_residual, float* df,
                       size_t hw, int B, hipStream_t s, int dt = HRN_F32);
int hrn_launch_fuse_scatter(const float* dsn, const float* dz, int n_in, int half, int pair_last, int alpha_residual, float* ds,
                            size_t hw, int B, hipStream_t s, int dt = HRN_F32);
// Decoder backward (HRNet.py:147-156,167-169): fused [N][H][W][64] f32, d_sr [N][3H][3W]; reference-layout parameters
// wd (64,64,3,3) = (Cin,Cout,kH,kW), bd (64), ad (1), wf (64), bf (1).  Writes d_fused; accumulates the five gradients.
int hrn_launch_decoder_bwd(const float* fused, const float* d_sr, const float* wd, const float* bd, const float* ad, const float* wf,
                           float* d_fused, float* dwd, float* dbd, float* dad, float* dwf, float* dbf, int N, int H, int W,
                           void* scratch, int num_cus, hipStream_t s);
size_t hrn_decoder_bwd_scratch_bytes(int num_cus);
// dx = conv3x3(g, W^T with taps flipped) (+ res): the data gradient of a cin -> cout convolution with raw weights
// w [cout][cin][3][3], on the forward f32 kernel.  wt / wtp: scratch for the transposed OIHW tensor and its packed form
// (cin*cout*9 floats each); zero_bias: max(cin, cout) zero floats.
int hrn_conv_dgrad(int cin, int cout, const float* w, const float* g, float* dx, const float* res, int M, int H, int W, float* wt,
                   void* wtp, const float* zero_bias, hipStream_t s, int dt = HRN_F32);
