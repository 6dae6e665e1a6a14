// Decoder backward (HRNet.py:147-156,167-169): sr = conv1x1(PReLU(ConvTranspose k3 s3 (fused))).
// Per LR pixel p and the 9 positions pos = (ky, kx) of its 3 x 3 output patch:
//     up[co][pos] = bd[co] + sum_ci F[p][ci] Wd[ci][co][pos];  y = PReLU(up);  sr[pos] = bf + sum_co wf[co] y[co][pos]
//     dy = wf[co] dsr[pos];  dup = dy PReLU'(up);  dF[p][ci] = sum_{co,pos} Wd[ci][co][pos] dup[co][pos]
//     dWd[ci][co][pos] += F[p][ci] dup[co][pos];  dbd[co] += sum_pos dup;  dad += sum dy min(up, 0);  dwf[co] += sum_pos y dsr;  dbf += sum dsr
// The forward never stores `up` (1.1 GiB at the bench size), so it is recomputed here.  One persistent 256-thread workgroup
// per CU: lane = co, wave = a quarter of the input channels (16 ci), whose 144 weights and 144 weight-gradient sums stay in
// registers for the whole launch.  Deterministic: per-workgroup partial slabs + a fixed-order finish.
#include "backward.h"

namespace {

constexpr int DB_DW = 64 * 64 * 9;
constexpr int DB_SLAB = DB_DW + 64 + 64 + 64 + 64;          // dWd | dbd[64] | dwf[64] | dad per lane[64] | dbf (+ pad)

__global__ __launch_bounds__(256, 1) void decoder_bwd_kernel(const float* __restrict__ fused, const float* __restrict__ d_sr,
                                                             const float* __restrict__ wd, const float* __restrict__ bd,
                                                             const float* __restrict__ ad, const float* __restrict__ wf,
                                                             float* __restrict__ d_fused, float* __restrict__ partial, int N, int H, int W) {
    __shared__ float upp[2][4][64][9];
    __shared__ float part[4][64][17];
    const int tid = threadIdx.x, co = tid & 63, cig = tid >> 6;
    float wreg[16][9], acc[16][9];
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int pos = 0; pos < 9; ++pos) {
            wreg[k][pos] = wd[((size_t)(16 * cig + k) * 64 + co) * 9 + pos];
            acc[k][pos] = 0.f;
        }
    const float bdc = bd[co], wfc = wf[co], a = ad[0];
    float a_dbd = 0.f, a_dwf = 0.f, a_dad = 0.f, a_dbf = 0.f;
    const long hw = (long)H * W, P = hw * N;
    const int W3 = 3 * W;
    int it = 0;
    for (long p = blockIdx.x; p < P; p += gridDim.x, ++it) {
        const long n = p / hw, rem = p - n * hw;
        const int y = (int)(rem / W), x = (int)(rem - (long)y * W);
        float f[16], ds[9], up[9], dup[9];
#pragma unroll
        for (int k = 0; k < 16; ++k) f[k] = fused[(size_t)p * 64 + 16 * cig + k];
        const float* dp = d_sr + ((size_t)n * 3 * H + 3 * y) * W3 + 3 * x;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) ds[ky * 3 + kx] = dp[(size_t)ky * W3 + kx];
#pragma unroll
        for (int pos = 0; pos < 9; ++pos) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += f[k] * wreg[k][pos];
            upp[it & 1][cig][co][pos] = s;
        }
        __syncthreads();
#pragma unroll
        for (int pos = 0; pos < 9; ++pos) {
            const float u = bdc + ((upp[it & 1][0][co][pos] + upp[it & 1][1][co][pos]) + (upp[it & 1][2][co][pos] + upp[it & 1][3][co][pos]));
            up[pos] = u;
            const float dy = wfc * ds[pos];
            dup[pos] = u > 0.f ? dy : a * dy;
            if (cig == 0) {
                a_dwf += (u > 0.f ? u : a * u) * ds[pos];
                a_dbd += dup[pos];
                a_dad += u > 0.f ? 0.f : dy * u;
                if (co == 0) a_dbf += ds[pos];
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float s = 0.f;
#pragma unroll
            for (int pos = 0; pos < 9; ++pos) {
                acc[k][pos] += f[k] * dup[pos];
                s += wreg[k][pos] * dup[pos];
            }
            part[cig][co][k] = s;
        }
        // dF[p][16 cig + k] = sum over co of part[cig][co][k]: lane (sub, k) sums 16 couts, then two cross-lane adds
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            const int k = co & 15, sub = co >> 4;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) s += part[cig][16 * sub + c][k];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (co < 16) d_fused[(size_t)p * 64 + 16 * cig + k] = s;
        }
        __builtin_amdgcn_wave_barrier();
    }
    float* out = partial + (size_t)blockIdx.x * DB_SLAB;
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int pos = 0; pos < 9; ++pos) out[((size_t)(16 * cig + k) * 64 + co) * 9 + pos] = acc[k][pos];
    if (cig == 0) {
        out[DB_DW + co] = a_dbd;
        out[DB_DW + 64 + co] = a_dwf;
        out[DB_DW + 128 + co] = a_dad;
        out[DB_DW + 192 + co] = co == 0 ? a_dbf : 0.f;
    }
}

// blocks 0 .. (DB_DW + 128) / 64 - 1: 64 consecutive elements x 16 slab phases; the last two blocks: the two scalars (slope, final
// bias), whose 64 per-lane terms per slab are summed by 64 x 16 threads as well.  Fixed order throughout.
__global__ __launch_bounds__(1024) void decoder_bwd_finish_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ dwd,
                                                                  float* __restrict__ dbd, float* __restrict__ dad, float* __restrict__ dwf,
                                                                  float* __restrict__ dbf) {
    __shared__ double red[16][64];
    const int el = threadIdx.x & 63, ph = threadIdx.x >> 6;
    constexpr int NEB = (DB_DW + 128) / 64;
    const bool scalar = (int)blockIdx.x >= NEB;
    const int idx = scalar ? (blockIdx.x == NEB ? DB_DW + 128 : DB_DW + 192) + el : blockIdx.x * 64 + el;
    double s = 0.0;
#pragma unroll 4
    for (int b = ph; b < nblk; b += 16) s += (double)partial[(size_t)b * DB_SLAB + idx];
    red[ph][el] = s;
    __syncthreads();
    if (ph != 0) return;
    s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][el];
    if (!scalar) {
        if (idx < DB_DW) dwd[idx] += (float)s;
        else if (idx < DB_DW + 64) dbd[idx - DB_DW] += (float)s;
        else dwf[idx - DB_DW - 64] += (float)s;
        return;
    }
    // the 64 lane terms of a scalar: one wave, fixed order
    red[0][el] = s;                                         // (one wave is left: its LDS accesses are ordered)
    if (el == 0) {
        double t = 0.0;
        for (int l = 0; l < 64; ++l) t += red[0][l];
        if ((int)blockIdx.x == NEB) dad[0] += (float)t;
        else dbf[0] += (float)t;
    }
}

}  // namespace

size_t hrn_decoder_bwd_scratch_bytes(int num_cus) { return (size_t)num_cus * DB_SLAB * 4; }

int hrn_launch_decoder_bwd(const float* fused, const float* d_sr, const float* wd, const float* bd, const float* ad, const float* wf,
                           float* d_fused, float* dwd, float* dbd, float* dad, float* dwf, float* dbf, int N, int H, int W,
                           void* scratch, int num_cus, hipStream_t s) {
    const long P = (long)N * H * W;
    int grid = num_cus;
    if (P < grid) grid = (int)P;
    hipLaunchKernelGGL(decoder_bwd_kernel, dim3(grid), dim3(256), 0, s, fused, d_sr, wd, bd, ad, wf, d_fused, (float*)scratch, N, H, W);
    static_assert((DB_DW + 128) % 64 == 0, "decoder_bwd_finish: 64 elements per block");
    hipLaunchKernelGGL(decoder_bwd_finish_kernel, dim3((DB_DW + 128) / 64 + 2), dim3(1024), 0, s, (const float*)scratch, grid, dwd, dbd,
                       dad, dwf, dbf);
    HRN_LAUNCH_CHECK();
    return 0;
}
