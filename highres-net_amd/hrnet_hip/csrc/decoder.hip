// Decoder: ConvTranspose2d(64,64,k=3,s=3) + PReLU + Conv2d(64->1, k=1), fused.
//   /root/reference/src/DeepNetworks/HRNet.py:147-156, :167-169
// stride == kernel, so every LR pixel produces its own 3x3 block of SR pixels:
//   sr[n, 3y+ky, 3x+kx] = bf + sum_co wf[co] * prelu( bd[co] + sum_ci s[n,y,x,ci] * Wd[ci,co,ky,kx] )
// i.e. nine 64x64 GEMMs per LR pixel followed by a 64-long dot product.  The reference materialises the
// (N,64,3H,3W) intermediate (1.1 GiB at B=32); here it never leaves the accumulator registers.
//
// Workgroup = 4 waves x 2 blocks of 32 consecutive LR pixels; the pixel operand (B) is loaded once from HBM straight
// into MFMA fragment layout and stays in registers; the nine weight slices stream through a double-buffered LDS
// stage exactly like the conv kernel's ("step" = 64 cout x 128 B of K).  D[co][pixel] orientation: each lane owns
// one pixel and 16 of the 32 output channels of a block, so the final 64->1 dot is 32 lane-local FMAs + one
// cross-half add.
#include "kernels.h"

namespace {

constexpr int W_ROW_PITCH = 144;
constexpr int W_BUF_BYTES = 64 * W_ROW_PITCH;

// SPLIT (DT = HRN_F32 only): the input is the bf16x3 pair of planes (hi at `fused`, lo `fused_lo` bytes further on); the pixel operand is
// formed as float(hi) + float(lo) on the way into the registers and everything after it is the fp32 decoder.
template <int DT, bool SPLIT>
__global__ __launch_bounds__(256, 2) void decoder_kernel(const void* __restrict__ fused, const void* __restrict__ wpk,
                                                         const float* __restrict__ bias, const float* __restrict__ slope,
                                                         const float* __restrict__ wf, const float* __restrict__ bfin,
                                                         float* __restrict__ sr, size_t npix, int H, int W, size_t fused_lo) {
    __shared__ __attribute__((aligned(16))) unsigned char w_lds[2 * W_BUF_BYTES];
    __shared__ __attribute__((aligned(16))) float bias_l[64];
    __shared__ __attribute__((aligned(16))) float wf_l[64];
    constexpr int ES = ElemOf<DT>::size;
    constexpr int NCHUNK = 64 * ES / 128;
    constexpr int NSTEP = 9 * NCHUNK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    if (tid < 64) { bias_l[tid] = bias[tid]; wf_l[tid] = wf[tid]; }

    const uint4* wg = (const uint4*)wpk;
    const int w_dst0 = (tid >> 3) * W_ROW_PITCH + (tid & 7) * 16;
    const int w_dst1 = w_dst0 + 32 * W_ROW_PITCH;
    {
        const uint4 w0 = wg[tid], w1 = wg[tid + 256];
        *(uint4*)(w_lds + w_dst0) = w0;
        *(uint4*)(w_lds + w_dst1) = w1;
    }

    // pixel operand: 2 blocks x (64 channels) in fragment layout, resident in registers
    size_t pixel[2];
    uint4 breg[2][NCHUNK][4];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        pixel[pb] = (size_t)blockIdx.x * 256 + wave * 64 + pb * 32 + r;
        const size_t pc = pixel[pb] < npix ? pixel[pb] : npix - 1;
        if constexpr (SPLIT) {
            // f32 fragment piece (c, k) = channels 32 c + 8 k + 4 hh .. + 3: 8 bytes of each bf16 plane
            const unsigned char* src = (const unsigned char*)fused + pc * 128 + hh * 8;
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint2 h = *(const uint2*)(src + c * 64 + k * 16), l = *(const uint2*)(src + fused_lo + c * 64 + k * 16);
                    uint4 v;
                    v.x = __float_as_uint(__uint_as_float(h.x << 16) + __uint_as_float(l.x << 16));
                    v.y = __float_as_uint(__uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u));
                    v.z = __float_as_uint(__uint_as_float(h.y << 16) + __uint_as_float(l.y << 16));
                    v.w = __float_as_uint(__uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u));
                    breg[pb][c][k] = v;
                }
        } else {
        const unsigned char* src = (const unsigned char*)fused + pc * 64 * ES + hh * 16;
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) breg[pb][c][k] = *(const uint4*)(src + c * 128 + k * 32);
        }
    }
    __syncthreads();

    const float a = slope[0];
    const float bfv = bfin[0];
    const unsigned char* a_base = w_lds + r * W_ROW_PITCH + hh * 16;
    float srv[2][9];
    int step = 0;
#pragma unroll
    for (int pos = 0; pos < 9; ++pos) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
        for (int chunk = 0; chunk < NCHUNK; ++chunk) {
            const bool more = step + 1 < NSTEP;
            uint4 w0, w1;
            if (more) {
                w0 = wg[(size_t)(step + 1) * 512 + tid];
                w1 = wg[(size_t)(step + 1) * 512 + tid + 256];
            }
            const unsigned char* wb = a_base + (step & 1) * W_BUF_BYTES;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (DT == HRN_BF16) {
                    const bf16x8 a0 = *(const bf16x8*)(wb + k * 32);
                    const bf16x8 a1 = *(const bf16x8*)(wb + 32 * W_ROW_PITCH + k * 32);
                    const bf16x8 b0 = __builtin_bit_cast(bf16x8, breg[0][chunk][k]);
                    const bf16x8 b1 = __builtin_bit_cast(bf16x8, breg[1][chunk][k]);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
                } else {
                    const f32x4 a0 = *(const f32x4*)(wb + k * 32);
                    const f32x4 a1 = *(const f32x4*)(wb + 32 * W_ROW_PITCH + k * 32);
                    const f32x4 b0 = __builtin_bit_cast(f32x4, breg[0][chunk][k]);
                    const f32x4 b1 = __builtin_bit_cast(f32x4, breg[1][chunk][k]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
                    }
                }
            }
            if (more) {
                unsigned char* wd = w_lds + ((step + 1) & 1) * W_BUF_BYTES;
                *(uint4*)(wd + w_dst0) = w0;
                *(uint4*)(wd + w_dst1) = w1;
            }
            __syncthreads();
            ++step;
        }
        // bias + PReLU + 64->1 projection for this sub-pixel position
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            float s = 0.f;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cb * 32 + 8 * g + 4 * hh;
                    const f32x4 bv = *(const f32x4*)(bias_l + co);
                    const f32x4 wv = *(const f32x4*)(wf_l + co);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = acc[pb][cb][4 * g + j] + bv[j];
                        t = t >= 0.f ? t : a * t;
                        s = fmaf(t, wv[j], s);
                    }
                }
            s += __shfl_xor(s, 32);
            srv[pb][pos] = s + bfv;
        }
    }
    if (hh == 0) {
        const size_t hw = (size_t)H * W;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            if (pixel[pb] >= npix) continue;
            const size_t n = pixel[pb] / hw, rem = pixel[pb] - n * hw;
            const int y = (int)(rem / W), x = (int)(rem - (size_t)y * W);
            float* dst = sr + n * hw * 9 + (size_t)(3 * y) * (3 * W) + 3 * x;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) dst[(size_t)ky * 3 * W + kx] = srv[pb][ky * 3 + kx];
        }
    }
}

// Wd (Cin=64, Cout=64, 3, 3) f32 -> [step = pos*NCHUNK + chunk][64 cout][128 B of cin]
template <int DT>
__global__ void decoder_pack_kernel(const float* __restrict__ w, void* __restrict__ out) {
    constexpr int ES = ElemOf<DT>::size;
    constexpr int KB = 128 / ES;
    constexpr int NCHUNK = 64 / KB;
    const int total = 64 * 64 * 9;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int kk = idx % KB;
        const int co = (idx / KB) % 64;
        const int step = idx / (KB * 64);
        const int chunk = step % NCHUNK, pos = step / NCHUNK;
        const int ci = chunk * KB + kk;
        store_elem<DT>(out, idx, w[(ci * 64 + co) * 9 + pos]);
    }
}

}  // namespace

int hrn_launch_decoder(int dt, const void* fused, const void* wpk, const float* bias, const float* slope,
                       const float* wf, const float* bf, float* sr, int N, int H, int W, hipStream_t stream, size_t fused_lo) {
    const size_t npix = (size_t)N * H * W;
    HRN_CHECK(npix > 0, -2, "decoder: empty input");
    const unsigned blocks = (unsigned)((npix + 255) / 256);
    HrnProfScope prof(dt == HRN_BF16 ? "decoder_bf16" : dt == HRN_BF16X3 ? "decoder_bf16x3" : "decoder_f32", (2.0 * 64 * 576 + 2.0 * 576) * npix,
                      (double)npix * (64.0 * hrn_esize(dt) + 36.0), stream);
    if (dt == HRN_BF16)
        hipLaunchKernelGGL((decoder_kernel<HRN_BF16, false>), dim3(blocks), dim3(256), 0, stream, fused, wpk, bias, slope, wf, bf, sr, npix, H, W, (size_t)0);
    else if (dt == HRN_BF16X3) {
        HRN_CHECK(fused_lo != 0, -2, "decoder bf16x3: lo-plane offset missing");
        hipLaunchKernelGGL((decoder_kernel<HRN_F32, true>), dim3(blocks), dim3(256), 0, stream, fused, wpk, bias, slope, wf, bf, sr, npix, H, W, fused_lo);
    } else
        hipLaunchKernelGGL((decoder_kernel<HRN_F32, false>), dim3(blocks), dim3(256), 0, stream, fused, wpk, bias, slope, wf, bf, sr, npix, H, W, (size_t)0);
    HRN_LAUNCH_CHECK();
    return 0;
}

int hrn_launch_decoder_pack(int dt, const float* w, void* packed, hipStream_t stream) {
    if (dt == HRN_BF16) hipLaunchKernelGGL(decoder_pack_kernel<HRN_BF16>, dim3(144), dim3(256), 0, stream, w, packed);
    else hipLaunchKernelGGL(decoder_pack_kernel<HRN_F32>, dim3(144), dim3(256), 0, stream, w, packed);
    HRN_LAUNCH_CHECK();
    return 0;
}
