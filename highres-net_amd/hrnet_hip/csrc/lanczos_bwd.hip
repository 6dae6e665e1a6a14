// Backward of the Lanczos-3 sub-pixel shift (src/lanczos.py:47-107; reached through ShiftNet.transform in apply_shifts,
// train.py:47-63, so that the registered loss trains both HRNet - through d img - and ShiftNet - through d shift).
//
// Forward, per plane (b, c): P = reflect-pad-3(img);  T[y][q] = sum_m ky[m] P[y+m][q];  out[y][x] = sum_n kx[n] T[y][x+n],
// ky = taps(shift[c][0]), kx = taps(shift[c][1]).
//   d img    d T[y][q] = sum_n kx[n] dout[y][q-n];  d P[p][q] = sum_m ky[m] d T[p-m][q];  d img = fold of d P back through the
//            reflection.  The fold is separable, so two 1-D adjoint passes do it: for a 1-D signal of length N and taps k,
//                g[i] = sum_m k[m] ( d[i-m+3] + [i > 0] d[3-m-i] + [i < N-1] d[2(N-1)-i-m+3] ),   terms with an index outside [0,N) dropped
//   d shift  dL/d kx[n] = sum dout[y][x] T[y][x+n];  dL/d ky[m] = sum dout[y][x] HP[y+m][x],  HP[p][x] = sum_n kx[n] P[p][x+n];
//            then through the tap function k_j = u_j / sum u, u = sinc(t) sinc(t/3), t = pi ((j-3) - d); where the forward
//            replaced t == 0 by 1e-6 the tap is a constant (torch.where, lanczos.py:33) and passes no gradient.
// Deterministic: per-tile partial sums + a fixed-order finish.
#include "kernels.h"

namespace {

constexpr int BT_H = 16, BT_W = 64;
constexpr int BH = BT_H + 6, BW = BT_W + 6;

__device__ __forceinline__ int reflect_b(int g, int n) {
    g = g < 0 ? -g : g;
    return g >= n ? 2 * (n - 1) - g : g;
}

__device__ __forceinline__ void taps7_and_grad(float d, float (&k)[7], float (&dk)[7]) {
    const float pi = 3.14159265358979323846f;
    float u[7], du[7], s = 0.f, ds = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const float x = (float)(j - 3) - d;
        float t = pi * x;
        const bool frozen = t == 0.f;
        t = frozen ? 1e-6f : t;
        const float t3 = t / 3.0f;
        const float A = sinf(t) / t, B = sinf(t3) / t3;
        const float dA = (cosf(t) * t - sinf(t)) / (t * t);
        const float dB = (cosf(t3) * t3 - sinf(t3)) / (t3 * t3) * (1.0f / 3.0f);
        u[j] = A * B;
        du[j] = frozen ? 0.f : -pi * (dA * B + A * dB);        // d u_j / d d  (t = pi (j - 3 - d))
        s += u[j];
        ds += du[j];
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        k[j] = u[j] / s;
        dk[j] = (du[j] * s - u[j] * ds) / (s * s);
    }
}

// ---- tap-gradient partial sums: partial[plane][tile][14] = (dL/dky[0..6], dL/dkx[0..6]) of this tile
__global__ __launch_bounds__(256) void lanczos_tapgrad_kernel(const float* __restrict__ img, const float* __restrict__ shift,
                                                              const float* __restrict__ dout, float* __restrict__ partial, int C, int H,
                                                              int W, int tiles) {
    __shared__ float tile[BH][BW + 1];
    __shared__ float vp[BT_H][BW + 1];          // T: vertical pass
    __shared__ float hp[BH][BT_W + 1];          // HP: horizontal pass of the padded rows
    __shared__ float kyx[2][7];
    __shared__ float red[14][256];
    const int plane = blockIdx.y, ch = plane % C;
    const int tiles_x = (W + BT_W - 1) / BT_W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * BT_H, x0 = tx * BT_W;
    const float* src = img + (size_t)plane * H * W;
    const float* dsrc = dout + (size_t)plane * H * W;
    if (threadIdx.x < 2) {
        float k[7], dk[7];
        taps7_and_grad(shift[ch * 2 + threadIdx.x], k, dk);
#pragma unroll
        for (int j = 0; j < 7; ++j) kyx[threadIdx.x][j] = k[j];
    }
    for (int i = threadIdx.x; i < BH * BW; i += 256) {
        const int yy = i / BW, xx = i - yy * BW;
        const int gy = reflect_b(y0 + yy - 3, H), gx = reflect_b(x0 + xx - 3, W);
        const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), cx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        tile[yy][xx] = src[(size_t)cy * W + cx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BT_H * BW; i += 256) {
        const int yy = i / BW, xx = i - yy * BW;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 7; ++m) s += kyx[0][m] * tile[yy + m][xx];
        vp[yy][xx] = s;
    }
    for (int i = threadIdx.x; i < BH * BT_W; i += 256) {
        const int yy = i / BT_W, xx = i - yy * BT_W;
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < 7; ++n) s += kyx[1][n] * tile[yy][xx + n];
        hp[yy][xx] = s;
    }
    __syncthreads();
    float acc[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) acc[k] = 0.f;
    for (int i = threadIdx.x; i < BT_H * BT_W; i += 256) {
        const int yy = i / BT_W, xx = i - yy * BT_W;
        const int gy = y0 + yy, gx = x0 + xx;
        if (gy < H && gx < W) {
            const float d = dsrc[(size_t)gy * W + gx];
#pragma unroll
            for (int m = 0; m < 7; ++m) acc[m] += d * hp[yy + m][xx];
#pragma unroll
            for (int n = 0; n < 7; ++n) acc[7 + n] += d * vp[yy][xx + n];
        }
    }
#pragma unroll
    for (int k = 0; k < 14; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int k = 0; k < 14; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x < 14) partial[((size_t)plane * tiles + blockIdx.x) * 14 + threadIdx.x] = red[threadIdx.x][0];
}

// d shift[c][0..1] += chain rule through the taps; one thread per (channel, axis)
__global__ void lanczos_shiftgrad_finish_kernel(const float* __restrict__ partial, const float* __restrict__ shift, int B, int C,
                                                int tiles, float* __restrict__ dshift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * C) return;
    const int c = i >> 1, axis = i & 1;
    double g[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < tiles; ++t) {
            const float* p = partial + (((size_t)b * C + c) * tiles + t) * 14 + axis * 7;
#pragma unroll
            for (int j = 0; j < 7; ++j) g[j] += (double)p[j];
        }
    float k[7], dk[7];
    taps7_and_grad(shift[c * 2 + axis], k, dk);
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 7; ++j) s += g[j] * (double)dk[j];
    dshift[c * 2 + axis] += (float)s;
}

// ---- 1-D adjoint passes.  AXIS 1: along x (taps of shift[c][1]); AXIS 0: along y (taps of shift[c][0]).
template <int AXIS>
__global__ __launch_bounds__(256) void lanczos_adjoint_kernel(const float* __restrict__ d, const float* __restrict__ shift,
                                                              float* __restrict__ g, int C, int H, int W) {
    __shared__ float k[7];
    const int plane = blockIdx.y, ch = plane % C;
    if (threadIdx.x == 0) {
        float kk[7], dk[7];
        taps7_and_grad(shift[ch * 2 + AXIS], kk, dk);
#pragma unroll
        for (int j = 0; j < 7; ++j) k[j] = kk[j];
    }
    __syncthreads();
    const size_t hw = (size_t)H * W;
    const float* src = d + (size_t)plane * hw;
    const int N = AXIS == 1 ? W : H;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < hw; idx += (size_t)gridDim.x * 256) {
        const int y = (int)(idx / W), x = (int)(idx - (size_t)y * W);
        const int i = AXIS == 1 ? x : y;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            float v = 0.f;
            const int a0 = i - m + 3, a1 = 3 - m - i, a2 = 2 * (N - 1) - i - m + 3;
            if ((unsigned)a0 < (unsigned)N) v += AXIS == 1 ? src[(size_t)y * W + a0] : src[(size_t)a0 * W + x];
            if (i > 0 && (unsigned)a1 < (unsigned)N) v += AXIS == 1 ? src[(size_t)y * W + a1] : src[(size_t)a1 * W + x];
            if (i < N - 1 && (unsigned)a2 < (unsigned)N) v += AXIS == 1 ? src[(size_t)y * W + a2] : src[(size_t)a2 * W + x];
            s += k[m] * v;
        }
        g[(size_t)plane * hw + idx] = s;
    }
}

}  // namespace

size_t hrn_lanczos_bwd_workspace_bytes_impl(int b, int c, int H, int W) {
    const size_t tiles = (size_t)((W + BT_W - 1) / BT_W) * ((H + BT_H - 1) / BT_H);
    return (size_t)b * c * H * W * 4 + (size_t)b * c * tiles * 14 * 4 + 256;
}

// d_img may be null (no image gradient wanted); d_shift [c][2] is accumulated (+=) and may be null
int hrn_launch_lanczos_shift_bwd(const float* img, const float* shift, const float* dout, int b, int c, int H, int W, float* d_img,
                                 float* d_shift, void* ws, hipStream_t s) {
    HRN_CHECK(H >= 4 && W >= 4, -2, "lanczos_shift backward: reflect padding of 3 needs H, W >= 4 (got %d x %d)", H, W);
    HRN_CHECK((long)b * c <= 65535, -2, "lanczos_shift backward: b*c = %ld exceeds the grid limit", (long)b * c);
    if (b * c == 0) return 0;
    const int tiles = ((W + BT_W - 1) / BT_W) * ((H + BT_H - 1) / BT_H);
    float* tmp = (float*)ws;
    float* partial = tmp + (size_t)b * c * H * W;
    if (d_shift) {
        hipLaunchKernelGGL(lanczos_tapgrad_kernel, dim3(tiles, b * c), dim3(256), 0, s, img, shift, dout, partial, c, H, W, tiles);
        hipLaunchKernelGGL(lanczos_shiftgrad_finish_kernel, dim3((2 * c + 63) / 64), dim3(64), 0, s, (const float*)partial, shift, b, c, tiles, d_shift);
    }
    if (d_img) {
        int gx = (int)(((size_t)H * W + 255) / 256);
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(lanczos_adjoint_kernel<1>, dim3(gx, b * c), dim3(256), 0, s, dout, shift, tmp, c, H, W);
        hipLaunchKernelGGL(lanczos_adjoint_kernel<0>, dim3(gx, b * c), dim3(256), 0, s, (const float*)tmp, shift, d_img, c, H, W);
    }
    HRN_LAUNCH_CHECK();
    return 0;
}
