// Lanczos-3 sub-pixel shift: tap synthesis + separable 7x7 gather with a 3-pixel reflect halo in LDS.
//   /root/reference/src/lanczos.py:5-43  (lanczos_kernel)   -> lanczos_taps_kernel / taps7()
//   /root/reference/src/lanczos.py:47-107 (lanczos_shift)   -> lanczos_shift_kernel
// The reference runs a Python loop over channels with ~10 framework launches per image (pad, 2x tap synthesis,
// 2x conv2d, crop, cat); here one launch handles all images, reading and writing every pixel exactly once
// (algorithmic traffic 2 * b*c*H*W * 4 B; HBM-bound).
//
// Semantics kept bit-for-bit in structure: fp32 taps k_j = sinc(pi x_j) * sinc(pi x_j / 3), x_j = (j-3) - d,
// pi*x == 0 -> 1e-6, no support window, normalised to sum 1; vertical pass first, then horizontal; reflect border
// without edge repeat.  (ReflectionPad2d(p>=3) + zero-padded conv + crop p == reflect-pad-3 + valid correlation.)
#include "kernels.h"

namespace {

__device__ __forceinline__ void taps7(float d, float (&k)[7]) {
    const float pi = 3.14159265358979323846f;     // np.pi rounded to f32, as torch does for tensor * python float
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const float x = (float)(j - 3) - d;
        float t = pi * x;
        t = (t == 0.f) ? 1e-6f : t;
        const float t3 = t / 3.0f;
        k[j] = (sinf(t) / t) * (sinf(t3) / t3);
        s += k[j];
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) k[j] = k[j] / s;
}

__global__ void lanczos_taps_kernel(const float* __restrict__ d, float* __restrict__ taps, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float k[7];
    taps7(d[i], k);
#pragma unroll
    for (int j = 0; j < 7; ++j) taps[i * 7 + j] = k[j];
}

constexpr int LT_H = 16, LT_W = 64;      // output tile; 256 threads
constexpr int LH = LT_H + 6, LW = LT_W + 6;

__device__ __forceinline__ int reflect(int g, int n) {
    g = g < 0 ? -g : g;
    return g >= n ? 2 * (n - 1) - g : g;
}

// img/out [b][c][H][W] f32; shift [c][2] = (dy, dx).  grid = (tiles_x * tiles_y, b * c)
__global__ __launch_bounds__(256) void lanczos_shift_kernel(const float* __restrict__ img, const float* __restrict__ shift,
                                                            float* __restrict__ out, int C, int H, int W) {
    __shared__ float tile[LH][LW + 1];
    __shared__ float tmp[LT_H][LW + 1];
    __shared__ float kyx[2][7];
    const int plane = blockIdx.y;             // b*C + c
    const int ch = plane % C;
    const int tiles_x = (W + LT_W - 1) / LT_W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * LT_H, x0 = tx * LT_W;
    const float* src = img + (size_t)plane * H * W;
    if (threadIdx.x < 2) {
        float k[7];
        taps7(shift[ch * 2 + threadIdx.x], k);
#pragma unroll
        for (int j = 0; j < 7; ++j) kyx[threadIdx.x][j] = k[j];
    }
    for (int i = threadIdx.x; i < LH * LW; i += 256) {
        const int yy = i / LW, xx = i - yy * LW;
        const int gy = reflect(y0 + yy - 3, H), gx = reflect(x0 + xx - 3, W);
        // tiles hanging over the image edge read clamped (unused) pixels
        const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), cx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        tile[yy][xx] = src[(size_t)cy * W + cx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LT_H * LW; i += 256) {      // vertical pass (lanczos.py:90)
        const int yy = i / LW, xx = i - yy * LW;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 7; ++m) s += kyx[0][m] * tile[yy + m][xx];
        tmp[yy][xx] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LT_H * LT_W; i += 256) {    // horizontal pass (lanczos.py:94)
        const int yy = i / LT_W, xx = i - yy * LT_W;
        const int gy = y0 + yy, gx = x0 + xx;
        if (gy < H && gx < W) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 7; ++m) s += kyx[1][m] * tmp[yy][xx + m];
            out[(size_t)plane * H * W + (size_t)gy * W + gx] = s;
        }
    }
}

}  // namespace

int hrn_launch_lanczos_taps(const float* d, int n, float* taps, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lanczos_taps_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, d, taps, n);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_lanczos_shift(const float* img, const float* shift, int b, int c, int H, int W, float* out, hipStream_t stream) {
    HRN_CHECK(H >= 4 && W >= 4, -2, "lanczos_shift: reflect padding of 3 needs H, W >= 4 (got %d x %d)", H, W);
    HRN_CHECK((long)b * c <= 65535, -2, "lanczos_shift: b*c = %ld exceeds the grid limit", (long)b * c);
    if (b * c == 0) return 0;
    const int tiles = ((W + LT_W - 1) / LT_W) * ((H + LT_H - 1) / LT_H);
    HrnProfScope prof("lanczos_shift", 28.0 * b * c * H * W, 8.0 * b * c * H * W, stream);
    hipLaunchKernelGGL(lanczos_shift_kernel, dim3(tiles, b * c), dim3(256), 0, stream, img, shift, out, c, H, W);
    HRN_LAUNCH_CHECK();
    return 0;
}
