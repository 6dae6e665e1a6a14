// Fused Adam over one flat fp32 buffer: `optimizer.step()` of src/train.py:191 (`optim.Adam(list(fusion_model.parameters()) +
// list(regis_model.parameters()), lr=...)`, train.py:252) as ONE launch for all 34.8 M parameters instead of ~10 small
// framework kernels per tensor.  The flat parameter / gradient buffers are the same ones the data-parallel gradient
// all-reduce works on (hrnet_hip/optim.py, hrnet_hip/dist.py).  Arithmetic = torch.optim.Adam (no amsgrad):
//     g += wd * p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//     p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound: 16 B read + 12 B written per parameter.
#include "../../../include/hrnet_hip.h"
#include "common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                                                   float step_size, float inv_bc2_sqrt) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 pv = ((f32x4*)p)[i], gv = ((const f32x4*)g)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = gv[j] + wd * pv[j];
            mv[j] = b1 * mv[j] + (1.f - b1) * gj;
            vv[j] = b2 * vv[j] + (1.f - b2) * gj * gj;
            pv[j] -= step_size * mv[j] / (sqrtf(vv[j]) * inv_bc2_sqrt + eps);
        }
        ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {             // tail (flat buffers are padded to 4 by the caller; kept for safety)
        const size_t i = n4 * 4 + threadIdx.x;
        const float gj = g[i] + wd * p[i];
        const float mj = b1 * m[i] + (1.f - b1) * gj, vj = b2 * v[i] + (1.f - b2) * gj * gj;
        m[i] = mj; v[i] = vj;
        p[i] -= step_size * mj / (sqrtf(vj) * inv_bc2_sqrt + eps);
    }
}

}  // namespace

extern "C" int hrn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, void* stream) {
    HRN_CHECK(params && grads && exp_avg && exp_avg_sq, -2, "hrn_adam_step: null argument");
    HRN_CHECK(step >= 1, -2, "hrn_adam_step: step must be >= 1 (got %d)", step);
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    size_t grid = (n / 4 + 255) / 256;
    if (grid < 1) grid = 1;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt);
    HRN_LAUNCH_CHECK();
    return 0;
}
