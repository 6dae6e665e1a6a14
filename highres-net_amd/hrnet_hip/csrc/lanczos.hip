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

constexpr int LT_H = 32, LT_W = 128;     // output tile; 256 threads.  (Round 1's 16 x 64 tile read 1.5 x its output and reached 1.5 TB/s at
constexpr int LH = LT_H + 6, LW = LT_W + 6;      // 1536 x 1536; this one reads 1.24 x, row by row per wave)
constexpr int LRG = 4;                   // rows per task of the vertical pass

__device__ __forceinline__ int reflect(int g, int n) {
    g = g < 0 ? -g : g;
    return g >= n ? 2 * (n - 1) - g : g;
}

// img/out [b][c][H][W] f32; shift [c][2] = (dy, dx).  grid = (tiles_x * tiles_y, b * c)
__global__ __launch_bounds__(256) void lanczos_shift_kernel(const float* __restrict__ img, const float* __restrict__ shift,
                                                            float* __restrict__ out, int C, int H, int W) {
    __shared__ float tile[LH][LW + 2];
    __shared__ float tmp[LT_H][LW + 2];
    __shared__ float kyx[2][7];
    const int plane = blockIdx.y;             // b*C + c
    const int ch = plane % C;
    const int tiles_x = (W + LT_W - 1) / LT_W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * LT_H, x0 = tx * LT_W;
    const float* src = img + (size_t)plane * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 2) {
        float k[7];
        taps7(shift[ch * 2 + threadIdx.x], k);
#pragma unroll
        for (int j = 0; j < 7; ++j) kyx[threadIdx.x][j] = k[j];
    }
    // halo tile, a row per wave at a time: consecutive lanes read consecutive pixels.  Reflected at the image border; tiles hanging over
    // the edge read clamped (unused) pixels
    // (all of a wave's ~30 loads are issued before the first is stored: 7.5 KB per wave in flight.  With a load - store pair per
    // iteration a CU had ~6 KB on the wire and the kernel sat at 1.5 TB/s whatever its tile)
    {
        constexpr int NR = (LH + 3) / 4, NC = (LW + 63) / 64;
        float v[NR][NC];
        int cxs[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int gx = reflect(x0 + lane + 64 * k - 3, W);
            cxs[k] = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int yy = wave + 4 * i;
            const int gy = reflect(y0 + (yy < LH ? yy : LH - 1) - 3, H);
            const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
            const float* row = src + (size_t)cy * W;
#pragma unroll
            for (int k = 0; k < NC; ++k) v[i][k] = (lane + 64 * k < LW) ? row[cxs[k]] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int yy = wave + 4 * i;
#pragma unroll
            for (int k = 0; k < NC; ++k)
                if (yy < LH && lane + 64 * k < LW) tile[yy][lane + 64 * k] = v[i][k];
        }
    }
    __syncthreads();
    // vertical pass (lanczos.py:90): a task = LRG output rows of one column from a sliding window of LRG + 6 values
    float ky[7], kx[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) { ky[m] = kyx[0][m]; kx[m] = kyx[1][m]; }
    for (int t = threadIdx.x; t < (LT_H / LRG) * LW; t += 256) {
        const int g = t / LW, xx = t - g * LW;
        float v[LRG + 6];
#pragma unroll
        for (int i = 0; i < LRG + 6; ++i) v[i] = tile[LRG * g + i][xx];
#pragma unroll
        for (int i = 0; i < LRG; ++i) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 7; ++m) s += ky[m] * v[i + m];
            tmp[LRG * g + i][xx] = s;
        }
    }
    __syncthreads();
    // horizontal pass (lanczos.py:94): thread = one column, every second row; a wave writes 256 contiguous bytes
    {
        const int xx = threadIdx.x & (LT_W - 1), gx = x0 + xx;
        for (int yy = threadIdx.x / LT_W; yy < LT_H; yy += 256 / LT_W) {
            const int gy = y0 + yy;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 7; ++m) s += kx[m] * tmp[yy][xx + m];
            if (gy < H && gx < W) out[(size_t)plane * H * W + (size_t)gy * W + gx] = s;
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
