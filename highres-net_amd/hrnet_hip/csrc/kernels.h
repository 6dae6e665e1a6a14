// Internal launcher declarations (one per kernel family).  All launch asynchronously on `stream`, allocate
// nothing and never synchronise (graph-capturable); they return 0 or a negative error with hrn_set_error() set.
#pragma once
#include "common.h"
#include "conv3x3.h"

// ---- stem.hip
int hrn_launch_median(const float* lrs, float* ref, int B, int V, int H, int W, hipStream_t stream);
int hrn_launch_stem(int dt, const float* in0, size_t img_stride0, const float* in1, int rep1, size_t img_stride1,
                    const float* sub, const float* w, const float* bias, const float* slope, void* out,
                    int M, int H, int W, hipStream_t stream, size_t out_lo = 0);       // out_lo: HRN_BF16X3's lo-plane byte offset
int hrn_launch_planes_to_f32(const void* hi, size_t lo_off, float* out, size_t n, hipStream_t stream);
int hrn_launch_f32_to_planes(const float* in, void* hi, size_t lo_off, size_t n, hipStream_t stream);
int hrn_launch_stem_pre(const float* in0, size_t img_stride0, const float* in1, int rep1, size_t img_stride1, const float* w,
                        const float* bias, float* out, int M, int H, int W, const float* only_if_nonpos, hipStream_t stream, int dt = HRN_F32);
int hrn_launch_plane_mean(const float* x, float* mean, int planes, size_t hw, hipStream_t stream);

// ---- decoder.hip
// fused [N][HW][64] (dt) -> sr [N][3H][3W] f32.  wpk: packed deconv weights (hrn_launch_decoder_pack), bias/slope/wf/bf f32.
int hrn_launch_decoder(int dt, const void* fused, const void* wpk, const float* bias, const float* slope,
                       const float* wf, const float* bf, float* sr, int N, int H, int W, hipStream_t stream, size_t fused_lo = 0);
int hrn_launch_decoder_pack(int dt, const float* w_iokk, void* packed, hipStream_t stream);

// ---- lanczos.hip
int hrn_launch_lanczos_taps(const float* d, int n, float* taps, hipStream_t stream);
int hrn_launch_lanczos_shift(const float* img, const float* shift, int b, int c, int H, int W, float* out, hipStream_t stream);
// ---- lanczos_bwd.hip: d_img (may be null) = adjoint of the shift, d_shift [c][2] (may be null) += gradient through the taps
size_t hrn_lanczos_bwd_workspace_bytes_impl(int b, int c, int H, int W);
int hrn_launch_lanczos_shift_bwd(const float* img, const float* shift, const float* dout, int b, int c, int H, int W, float* d_img,
                                 float* d_shift, void* ws, hipStream_t stream);

// ---- losses.hip
int hrn_launch_masked_cmse(const float* srs, const float* hrs, const float* maps, int B, int S, int crop, int metric, float* out,
                           hipStream_t stream);
// differentiable registered-loss tail (train.py:78-87, :183-187): forward keeps stats[B][4] = {n, bias, cMSE, 0} for the backward
size_t hrn_loss_train_workspace_bytes_impl(int B);
int hrn_launch_loss_train(const float* srs, const float* hrs, const float* maps, int B, int S, int crop, int metric, float* out,
                          double* stats, double* partial, hipStream_t stream);
int hrn_launch_loss_backward(const float* srs, const float* hrs, const float* maps, const double* stats, const float* d_out, int B,
                             int S, int crop, int metric, float* d_srs, hipStream_t stream);
int hrn_launch_shift_cpsnr(const float* srs, const float* hrs, const float* maps, int B, int S, int border, int clip,
                           double* scores, float* out, hipStream_t stream);

// ---- shiftnet.hip
int hrn_launch_bn_stats(const float* x, size_t npix, int C, const float* gamma, const float* beta, float eps,
                        float* scale, float* shift, float* running_mean, float* running_var, float momentum,
                        double* partial, int partial_blocks, hipStream_t stream);
int hrn_launch_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                       const float* conv_bias, float* scale, float* shift, int C, hipStream_t stream);
int hrn_launch_bn_act_pool(const float* x, const float* scale, const float* shift, float* out, int N, int H, int W, int C,
                           int pool, hipStream_t stream);
// fc1: xr = the input in the reference's flatten order (hrn_launch_fc_to_ref: from the NHWC activation, dropout folded in), w = the raw
// fc1.weight (1024, 32768), partial = hrn_fc1_partial_bytes() of scratch
int hrn_launch_fc_to_ref(const float* y, const unsigned char* mask, float* xr, int B, hipStream_t stream);
size_t hrn_fc1_partial_bytes(void);
int hrn_launch_fc1(const float* xr, const float* w, const float* b, float* y, int B, float* partial, hipStream_t stream);
int hrn_launch_fc2(const float* y, const float* w, float* theta, int B, hipStream_t stream);
