/* hrnet_hip.h - C ABI of libhrnet_hip.so: the MI355X (gfx950) implementation of the HighRes-net hot path.
 *
 * The reference (gwall-ceres/HighRes-net) has no FFI: its boundary is three Python symbols.  Every entry point
 * below replaces one of them (or one stage of one); `highres-net_amd/hrnet_hip/binding.py` binds them with
 * ctypes and `highres-net_amd/{DeepNetworks/HRNet.py,DeepNetworks/ShiftNet.py,lanczos.py}` re-expose the
 * reference's module names on top.
 *
 *   hrn_hrnet_forward      <-  HRNet.forward(lrs, alphas)            src/DeepNetworks/HRNet.py:186-211
 *   hrn_encoder_forward    <-  median/stack + Encoder.forward         src/DeepNetworks/HRNet.py:200-206, :62-74
 *   hrn_fuse_forward       <-  RecuversiveNet.forward                 src/DeepNetworks/HRNet.py:99-134
 *   hrn_decoder_forward    <-  Decoder.forward                        src/DeepNetworks/HRNet.py:158-169
 *   hrn_shiftnet_forward   <-  ShiftNet.forward(x)                    src/DeepNetworks/ShiftNet.py:49-75
 *   hrn_lanczos_shift      <-  lanczos.lanczos_shift(img, shift, ...) src/lanczos.py:47-107
 *                              (and ShiftNet.transform, ShiftNet.py:77-90, which only re-labels its arguments)
 *   hrn_lanczos_kernel     <-  lanczos.lanczos_kernel(dx, a=3, N=7)   src/lanczos.py:5-43
 *   hrn_*_pack             <-  nn.Module.load_state_dict / .to(device): reference-layout f32 parameters
 *                              (OIHW conv, (Cin,Cout,kH,kW) deconv, (out,in) linear) -> kernel layouts
 *   hrn_hrnet_forward_train / hrn_hrnet_backward, hrn_shiftnet_forward_train / hrn_shiftnet_backward,
 *   hrn_lanczos_shift_backward  <-  torch autograd through the three symbols above, src/train.py:172-190
 *   hrn_adam_step          <-  optimizer.step() of torch.optim.Adam   src/train.py:191, :252
 *   hrn_get_loss / hrn_shift_cpsnr  <-  get_loss (train.py:66-87) / shift_cPSNR (Evaluator.py:52-73)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed or a torch CUDA tensor's data_ptr) unless stated;
 *     all tensors are dense / contiguous; images are square-agnostic here (the Python mirror asserts H == W
 *     where the reference's .view() does, HRNet.py:204);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls only enqueue work: no allocation,
 *     no synchronisation, no host read-back, so a call sequence can be captured into a hipGraph;
 *   - the caller owns every buffer, including the workspace (size from the matching *_workspace_bytes);
 *   - return value: 0 on success, negative on error (-2 bad argument, -3 workspace/packed buffer too small,
 *     -5 HIP runtime error); hrn_last_error() returns a thread-local message for the last failing call;
 *   - dtype selects storage of activations and the MFMA input type:
 *       HRN_DTYPE_F32  : f32 activations, v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation)
 *       HRN_DTYPE_BF16 : bf16 activations/weights, v_mfma_f32_32x32x16_bf16, fp32 accumulation
 *       HRN_DTYPE_BF16X3 : every fp32 activation / weight as two bf16 planes (hi = bf16(v), lo = bf16(v - hi)); a product is
 *                        hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (~2^-16 per product): the
 *                        reference's fp32 arithmetic (train.py:168-171, predict.py:36-37) to ~1e-5 at a third of the bf16
 *                        matrix rate.  Stage tensors (emb, fused) are [2 planes][...][64] bf16, lo plane directly behind hi.
 *     inputs (lrs, alphas, ShiftNet pairs, Lanczos images) and the SR output are always f32.
 */
#ifndef HRNET_HIP_H
#define HRNET_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HRN_DTYPE_F32 0
#define HRN_DTYPE_BF16 1
#define HRN_DTYPE_BF16X3 2
#define HRN_MAX_RES_LAYERS 8
#define HRN_ABI_VERSION 1

int hrn_version(void);
const char* hrn_last_error(void);

/* ------------------------------------------------------------------ HRNet */
/* Parameters exactly as the reference's state_dict holds them (f32, contiguous). */
typedef struct hrn_hrnet_params {
    int num_layers;                                   /* config["encoder"]["num_layers"], 0..HRN_MAX_RES_LAYERS */
    const float* enc_init_w;                          /* encode.init_layer.0.weight (64,2,3,3) */
    const float* enc_init_b;                          /* encode.init_layer.0.bias   (64)       */
    const float* enc_init_a;                          /* encode.init_layer.1.weight (1)  PReLU */
    const float* enc_res_w[2 * HRN_MAX_RES_LAYERS];   /* encode.res_layers.L.block.{0,2}.weight (64,64,3,3), index 2L+{0,1} */
    const float* enc_res_b[2 * HRN_MAX_RES_LAYERS];   /* ... .bias (64) */
    const float* enc_res_a[2 * HRN_MAX_RES_LAYERS];   /* encode.res_layers.L.block.{1,3}.weight (1) */
    const float* enc_final_w;                         /* encode.final.0.weight (64,64,3,3) */
    const float* enc_final_b;                         /* encode.final.0.bias */
    const float* fuse_res_w[2];                       /* fuse.fuse.0.block.{0,2}.weight (128,128,3,3) */
    const float* fuse_res_b[2];
    const float* fuse_res_a[2];                       /* fuse.fuse.0.block.{1,3}.weight (1) */
    const float* fuse_out_w;                          /* fuse.fuse.1.weight (64,128,3,3) */
    const float* fuse_out_b;
    const float* fuse_out_a;                          /* fuse.fuse.2.weight (1) */
    const float* dec_w;                               /* decode.deconv.0.weight (64,64,3,3) = (Cin,Cout,kH,kW) */
    const float* dec_b;
    const float* dec_a;                               /* decode.deconv.1.weight (1) */
    const float* fin_w;                               /* decode.final.weight (1,64,1,1) */
    const float* fin_b;                               /* decode.final.bias (1) */
} hrn_hrnet_params;

size_t hrn_hrnet_packed_bytes(int dtype, int num_layers);
int hrn_hrnet_pack(const hrn_hrnet_params* params, int dtype, void* packed, size_t packed_bytes, void* stream);

size_t hrn_hrnet_workspace_bytes(int dtype, int B, int V, int H, int W);

/* lrs (B,V,H,W) f32, alphas (B,V) f32 -> sr (B,1,3H,3W) f32 */
int hrn_hrnet_forward(const void* packed, int dtype, int num_layers, int alpha_residual,
                      const float* lrs, const float* alphas, int B, int V, int H, int W,
                      float* sr, void* workspace, size_t workspace_bytes, void* stream);

/* Stages (same workspace).  emb: view stack [B][V][H][W][64] in `dtype` (channels-last); fused: [B][H][W][64].
 * HRN_DTYPE_BF16X3: emb is [2][B][V][H][W][64] bf16 and fused [2][B][H][W][64] bf16 (plane 0 = hi, plane 1 = lo). */
int hrn_encoder_forward(const void* packed, int dtype, int num_layers, const float* lrs, int B, int V, int H, int W,
                        void* emb, void* workspace, size_t workspace_bytes, void* stream);
/* Destroys `emb` (levels are reduced in place, HRNet.py:113-132). */
int hrn_fuse_forward(const void* packed, int dtype, int num_layers, int alpha_residual, void* emb, const float* alphas,
                     int B, int V, int H, int W, void* fused, void* workspace, size_t workspace_bytes, void* stream);
int hrn_decoder_forward(const void* packed, int dtype, int num_layers, const void* fused, int N, int H, int W,
                        float* sr, void* stream);

/* Training path (fp32 only): `srs = fusion_model(lrs, alphas)` with grad enabled and `loss.backward()` through HRNet,
 * src/train.py:172-190.  hrn_hrnet_forward_train is hrn_hrnet_forward(HRN_DTYPE_F32) with every intermediate kept in
 * `train_ws`; hrn_hrnet_backward consumes that workspace (same B, V, H, W) and d_sr = dLoss/d sr (B,1,3H,3W) and
 * ACCUMULATES (+=, like autograd's .grad) the parameter gradients into the buffers `grads` points at - the same struct,
 * fields aliasing f32 gradient tensors of the parameters' shapes (the inputs lrs / alphas get no gradient, as in
 * train.py).  `packed` is the HRN_DTYPE_F32 blob of hrn_hrnet_pack, `params` the raw reference-layout tensors.
 * Any PReLU slope is accepted, as in the reference: with a positive slope the backward works from the stored post-activations;
 * behind a slope <= 0 it recomputes the pre-activation (decided on the device: the extra launches exit at once otherwise). */
size_t hrn_hrnet_train_workspace_bytes(int num_layers, int B, int V, int H, int W);
int hrn_hrnet_forward_train(const void* packed, int num_layers, int alpha_residual, const float* lrs, const float* alphas,
                            int B, int V, int H, int W, float* sr, void* train_ws, size_t train_ws_bytes, void* stream);
int hrn_hrnet_backward(const void* packed, const hrn_hrnet_params* params, int alpha_residual, const float* lrs,
                       const float* alphas, int B, int V, int H, int W, const float* d_sr, const hrn_hrnet_params* grads,
                       void* train_ws, size_t train_ws_bytes, void* stream);
/* The same with a dtype: HRN_DTYPE_F32 (what the two entry points above run) or HRN_DTYPE_BF16X3 - every activation and gradient
 * tensor of the workspace a pair of bf16 planes, three bf16 MFMAs per product in the convolutions, their data gradients and their
 * weight gradients (same workspace size; `packed` is then the HRN_DTYPE_BF16X3 blob).  Parameters, gradients, lrs, sr, d_sr: f32. */
int hrn_hrnet_forward_train_dt(const void* packed, int dtype, int num_layers, int alpha_residual, const float* lrs,
                               const float* alphas, int B, int V, int H, int W, float* sr, void* train_ws, size_t train_ws_bytes,
                               void* stream);
int hrn_hrnet_backward_dt(const void* packed, int dtype, const hrn_hrnet_params* params, int alpha_residual, const float* lrs,
                          const float* alphas, int B, int V, int H, int W, const float* d_sr, const hrn_hrnet_params* grads,
                          void* train_ws, size_t train_ws_bytes, void* stream);

/* ------------------------------------------------------------------ ShiftNet */
typedef struct hrn_shiftnet_params {
    const float* conv_w[8];       /* layerN.0.weight  (co,ci,3,3): 2->64,64->64 x3,64->128,128->128 x3 */
    const float* conv_b[8];       /* layerN.0.bias */
    const float* bn_g[8];         /* layerN.1.weight */
    const float* bn_b[8];         /* layerN.1.bias */
    float* bn_rm[8];              /* layerN.1.running_mean (updated in place when train_bn != 0) */
    float* bn_rv[8];              /* layerN.1.running_var  (updated in place when train_bn != 0) */
    const float* fc1_w;           /* fc1.weight (1024, 32768), reference flatten order c*256 + h*16 + w */
    const float* fc1_b;           /* fc1.bias (1024) */
    const float* fc2_w;           /* fc2.weight (2, 1024) */
} hrn_shiftnet_params;

size_t hrn_shiftnet_packed_bytes(void);
int hrn_shiftnet_pack(const hrn_shiftnet_params* params, void* packed, size_t packed_bytes, void* stream);
size_t hrn_shiftnet_workspace_bytes(int B);

/* x (B,2,128,128) f32 -> theta (B,2) f32.  `params` supplies the live BatchNorm tensors (affine + running stats) and
 * fc1_w, which is read IN PLACE in the reference's layout (its 134 MB are not part of `packed`); the other conv/fc
 * pointers are not read here (they live in `packed`).
 * train_bn != 0: batch statistics (biased var) and running-stat update with `momentum` (nn.BatchNorm2d train mode);
 * dropout_mask: NULL (eval) or uint8 (B,32768) keep-mask in the reference's flatten order; kept activations x2. */
int hrn_shiftnet_forward(const void* packed, const hrn_shiftnet_params* params, const float* x, int B,
                         int train_bn, float momentum, const unsigned char* dropout_mask, float* theta,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Training path (ShiftNet in .train() mode; `shifts = register_batch(regis_model, ...)` ... `loss.backward()`,
 * src/train.py:176-190).  hrn_shiftnet_forward_train = hrn_shiftnet_forward(train_bn = 1) with every layer's tensors
 * and batch statistics kept in `train_ws`; hrn_shiftnet_backward turns d_theta (B,2) into the parameter gradients,
 * ACCUMULATED (+=) into the buffers of `grads` (conv_w/conv_b/bn_g/bn_b/fc1_w/fc1_b/fc2_w in the parameters' own
 * reference layouts; bn_rm/bn_rv are not read), and, when d_x is not NULL, writes the gradient of the input pairs
 * d_x (B,2,128,128).  `params` are the raw reference-layout tensors, `dropout_mask` the mask the forward used.  B <= 32. */
size_t hrn_shiftnet_train_workspace_bytes(int B);
int hrn_shiftnet_forward_train(const void* packed, const hrn_shiftnet_params* params, const float* x, int B, float momentum,
                               const unsigned char* dropout_mask, float* theta, void* train_ws, size_t train_ws_bytes,
                               void* stream);
int hrn_shiftnet_backward(const hrn_shiftnet_params* params, const float* x, int B, const unsigned char* dropout_mask,
                          const float* d_theta, const hrn_shiftnet_params* grads, float* d_x, void* train_ws,
                          size_t train_ws_bytes, void* stream);

/* ------------------------------------------------------------------ Lanczos */
/* dx (n) f32 -> taps (n,7) f32;  a = 3, N = 7 (the only values the reference's call sites use). */
int hrn_lanczos_kernel(const float* dx, int n, float* taps, void* stream);
/* img (b,c,H,W) f32, shift (c,2) = (dy,dx) per channel -> out (b,c,H,W) f32;  a = 3, N = 7, any p >= 3. */
int hrn_lanczos_shift(const float* img, const float* shift, int b, int c, int H, int W, float* out, void* stream);
/* Backward of hrn_lanczos_shift (autograd through lanczos.lanczos_shift / ShiftNet.transform in apply_shifts,
 * src/train.py:47-63, lanczos.py:47-107): d_img (b,c,H,W) = adjoint of the shift applied to d_out (may be NULL),
 * d_shift (c,2) += gradient through the Lanczos taps, summed over b (may be NULL). */
size_t hrn_lanczos_shift_backward_workspace_bytes(int b, int c, int H, int W);
int hrn_lanczos_shift_backward(const float* img, const float* shift, const float* d_out, int b, int c, int H, int W,
                               float* d_img, float* d_shift, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ loss / score reductions
 * hrn_get_loss     <-  get_loss(srs, hrs, hr_maps, metric)   src/train.py:66-87, with get_crop_mask (:90-106) folded in:
 *                      srs/hrs/hr_maps (B,S,S) f32 -> out (B) f32.  metric: 0 'masked_MSE', 1 'cMSE', 2 'cPSNR' (returns
 *                      -10 log10(cMSE) exactly like the reference); crop: border width forced to mask 0 (0 = none).
 * hrn_shift_cpsnr  <-  shift_cPSNR(np.clip(sr,0,1), hr, hr_map, border_w)   src/Evaluator.py:52-73 (cPSNR :11-43),
 *                      batched: (B,S,S) f32 -> out (B) f32 = max over the (2 border + 1)^2 integer offsets of hr;
 *                      clip != 0 clamps sr to [0,1] first (predict.py:43, train.py:212); workspace: B*(2b+1)^2 doubles.
 *                      Status maps are binary (Evaluator's formula squares the mask; identical for 0/1 maps). */
int hrn_get_loss(const float* srs, const float* hrs, const float* hr_maps, int B, int S, int crop, int metric, float* out, void* stream);
/* The same loss as the differentiable tail of a training step (src/train.py:183-187: loss = -get_loss(srs_shifted, hrs,
 * mask, 'cPSNR')): hrn_get_loss_train also writes stats (B,4) f64 = {n, brightness bias b, cMSE, 0} per sample;
 * hrn_get_loss_backward turns d_out (B) into d_srs (B,S,S) with b held constant, as the reference detaches it (train.py:83):
 * d cMSE / d sr = 2 m (sr + b - hr) / n.  metric: 1 'cMSE' or 2 'cPSNR'.  workspace: fp64 partial sums (fixed-order: reproducible). */
size_t hrn_get_loss_train_workspace_bytes(int B);
int hrn_get_loss_train(const float* srs, const float* hrs, const float* hr_maps, int B, int S, int crop, int metric, float* out,
                       double* stats, void* workspace, size_t workspace_bytes, void* stream);
int hrn_get_loss_backward(const float* srs, const float* hrs, const float* hr_maps, const double* stats, const float* d_out, int B,
                          int S, int crop, int metric, float* d_srs, void* stream);
size_t hrn_shift_cpsnr_workspace_bytes(int B, int border);
int hrn_shift_cpsnr(const float* srs, const float* hrs, const float* hr_maps, int B, int S, int border, int clip, float* out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ optimiser
 * hrn_adam_step  <-  optimizer.step() of torch.optim.Adam (src/train.py:191, :252), one launch over a flat fp32 buffer
 *                    holding every parameter of both models (the buffer the gradient all-reduce also works on):
 *                    g += weight_decay p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *                    p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps).  step counts from 1. */
int hrn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void* stream);

/* ------------------------------------------------------------------ built-in kernel timing (hipEvent pairs)
 * The reference has no profiling hooks; these exist so that bench.py can state, live, the
 * achieved TFLOP/s / GB/s of each kernel family against the gfx950 roofline.  enable(1) clears the table and starts
 * recording on every subsequent launch; enable(0) stops.  get() synchronises the recorded events (host blocks). */
int hrn_profile_enable(int on);
int hrn_profile_count(void);
int hrn_profile_get(int idx, char* name, int name_len, long* launches, double* total_ms, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* HRNET_HIP_H */
