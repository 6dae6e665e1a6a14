// Shared device/host helpers for the gfx950 HighRes-net kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "prof.h"

#define HRN_F32 0
#define HRN_BF16 1
#define HRN_BF16X3 2     // fp32 values as two bf16 planes (hi, lo); three bf16 MFMAs per product (conv3x3_v6x3.hip)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // 16-byte staging unit (native vector: stays in VGPRs)
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;   // 8 bytes = 4 bf16

// ------------------------------------------------------------------ error plumbing (host)
void hrn_set_error(const char* fmt, ...);
// allow `kernel` to use `bytes` of dynamic LDS on the current device (set once per device; prof.hip)
int hrn_allow_lds(const void* kernel, int bytes);
// compute units of the current device, cached per device (prof.hip)
int hrn_device_cus(void);
#define HRN_CHECK(cond, code, ...)                                \
    do {                                                          \
        if (!(cond)) { hrn_set_error(__VA_ARGS__); return (code); } \
    } while (0)
#define HRN_HIP(call)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            hrn_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -5;                                                                   \
        }                                                                                \
    } while (0)
#define HRN_LAUNCH_CHECK()  HRN_HIP(hipGetLastError())

// ------------------------------------------------------------------ bf16 helpers (device)
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned)b) << 16);
}
// round-to-nearest-even via the hardware convert (v_cvt_pk_bf16_f32 on gfx950, NaN-safe)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}
// two floats -> one dword of two bf16 (lo in bits 0..15): ONE v_cvt_pk_bf16_f32.  (Converting the halves separately and
// or-ing them costs four instructions; the epilogues do this for every output pair.)
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// two floats -> their (hi, lo) bf16 split, packed like pack2_bf16: hi = bf16(v), lo = bf16(v - hi)
__device__ __forceinline__ void split2_bf16(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pack2_bf16(a, b);
    lo = pack2_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

// 4 x 4 transpose of 16-byte items across each quad of lanes (4k .. 4k+3): x[j] of lane i <- x[i] of lane j (b0 / b1 = bit 0 /
// bit 1 of the lane id).  Two butterfly stages of v_mov_dpp quad_perm + v_cndmask per dword, 64 VALU per call.  The conv
// epilogues use it to turn "a lane owns 64 contiguous bytes of its own pixel" (one store instruction = 64 pieces of 16 bytes
// in 32 different 128-byte lines) into "the four lanes of a quad own the four pieces of one pixel" (8 whole lines per store
// instruction, together with the other half-wave) where they have the registers and the VALU time for it.
__device__ __forceinline__ void quad_transpose(u32x4 (&x)[4], bool b0, bool b1) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int p = 0; p < 4; p += 2) {
            const unsigned a = x[p][d], b = x[p + 1][d];
            const unsigned ra = (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
            const unsigned rb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xF, 0xF, true);
            x[p][d] = b0 ? ra : a;
            x[p + 1][d] = b0 ? b : rb;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const unsigned a = x[p][d], b = x[p + 2][d];
            const unsigned ra = (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
            const unsigned rb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xF, 0xF, true);
            x[p][d] = b1 ? ra : a;
            x[p + 2][d] = b1 ? b : rb;
        }
    }
}

template <int DT> struct ElemOf;
template <> struct ElemOf<HRN_F32>  { typedef float type;          static constexpr int size = 4; };
template <> struct ElemOf<HRN_BF16> { typedef unsigned short type; static constexpr int size = 2; };

template <int DT> __device__ __forceinline__ float load_elem(const void* p, size_t i);
template <> __device__ __forceinline__ float load_elem<HRN_F32>(const void* p, size_t i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float load_elem<HRN_BF16>(const void* p, size_t i) { return bf16_bits_to_f32(((const unsigned short*)p)[i]); }

template <int DT> __device__ __forceinline__ void store_elem(void* p, size_t i, float v);
template <> __device__ __forceinline__ void store_elem<HRN_F32>(void* p, size_t i, float v) { ((float*)p)[i] = v; }
template <> __device__ __forceinline__ void store_elem<HRN_BF16>(void* p, size_t i, float v) { ((unsigned short*)p)[i] = f32_to_bf16_bits(v); }

// 4 consecutive elements (16-byte aligned for f32, 8-byte aligned for bf16)
template <int DT> __device__ __forceinline__ f32x4 load4(const void* p, size_t i);
template <> __device__ __forceinline__ f32x4 load4<HRN_F32>(const void* p, size_t i) {
    return *(const f32x4*)((const float*)p + i);
}
template <> __device__ __forceinline__ f32x4 load4<HRN_BF16>(const void* p, size_t i) {
    uint2 u = *(const uint2*)((const unsigned short*)p + i);
    f32x4 r;
    r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xffff0000u);
    r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xffff0000u);
    return r;
}
template <int DT> __device__ __forceinline__ void store4(void* p, size_t i, f32x4 v);
template <> __device__ __forceinline__ void store4<HRN_F32>(void* p, size_t i, f32x4 v) {
    *(f32x4*)((float*)p + i) = v;
}
template <> __device__ __forceinline__ void store4<HRN_BF16>(void* p, size_t i, f32x4 v) {
    uint2 u; u.x = pack2_bf16(v[0], v[1]); u.y = pack2_bf16(v[2], v[3]);
    *(uint2*)((unsigned short*)p + i) = u;
}

// ------------------------------------------------------------------ activation tensors of the training path
// f32, or (X3 = the bf16x3 mode) a PAIR of bf16 planes: hi at p, lo `lo` bytes further on; a tensor of n elements has lo = 2 n (the lo
// plane directly behind the hi plane), which is how every kernel derives it from the tensor's shape.  Unit: 4 consecutive elements.
template <bool X3> __device__ __forceinline__ f32x4 act_ld4(const void* p, size_t lo, size_t i4) {
    if constexpr (X3) {
        const u32x2 h = __builtin_nontemporal_load((const u32x2*)p + i4);
        const u32x2 l = __builtin_nontemporal_load((const u32x2*)((const unsigned char*)p + lo) + i4);
        f32x4 r;
        r[0] = __uint_as_float(h[0] << 16) + __uint_as_float(l[0] << 16);
        r[1] = __uint_as_float(h[0] & 0xffff0000u) + __uint_as_float(l[0] & 0xffff0000u);
        r[2] = __uint_as_float(h[1] << 16) + __uint_as_float(l[1] << 16);
        r[3] = __uint_as_float(h[1] & 0xffff0000u) + __uint_as_float(l[1] & 0xffff0000u);
        return r;
    } else {
        return __builtin_nontemporal_load((const f32x4*)p + i4);
    }
}
template <bool X3> __device__ __forceinline__ void act_st4(void* p, size_t lo, size_t i4, f32x4 v) {
    if constexpr (X3) {
        unsigned h0, l0, h1, l1;
        split2_bf16(v[0], v[1], h0, l0);
        split2_bf16(v[2], v[3], h1, l1);
        const u32x2 h = {h0, h1}, l = {l0, l1};
        __builtin_nontemporal_store(h, (u32x2*)p + i4);
        __builtin_nontemporal_store(l, (u32x2*)((unsigned char*)p + lo) + i4);
    } else {
        __builtin_nontemporal_store(v, (f32x4*)p + i4);
    }
}
// one element
template <bool X3> __device__ __forceinline__ float act_ld1(const void* p, size_t lo, size_t i) {
    if constexpr (X3) return bf16_bits_to_f32(((const unsigned short*)p)[i]) + bf16_bits_to_f32(((const unsigned short*)((const unsigned char*)p + lo))[i]);
    else return ((const float*)p)[i];
}

// (s, ss) = sum over k < nblk of partial[(k * C + c) * 2 + {0, 1}] for channel c = threadIdx.x % C, by a block of 1024 threads = C channels
// x (1024 / C) phases (C = 64 or 128), fixed order; valid in the threads with threadIdx.x < C.  A finish kernel is a chain of dependent
// loads: one thread per channel took 65 us for 256 slabs, this takes ~5.
__device__ __forceinline__ void block_pair_sum(const double* __restrict__ partial, int nblk, int C, double& s, double& ss) {
    __shared__ double red_ps[2][1024];
    const int c = (int)threadIdx.x % C, ph = (int)threadIdx.x / C, nph = (int)blockDim.x / C;
    double a = 0.0, b = 0.0;
#pragma unroll 4
    for (int k = ph; k < nblk; k += nph) { a += partial[((size_t)k * C + c) * 2]; b += partial[((size_t)k * C + c) * 2 + 1]; }
    red_ps[0][threadIdx.x] = a; red_ps[1][threadIdx.x] = b;
    __syncthreads();
    if (ph == 0)
        for (int k = 1; k < nph; ++k) { a += red_ps[0][k * C + c]; b += red_ps[1][k * C + c]; }
    s = a; ss = b;
}

static inline size_t hrn_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// bytes per element of an activation / weight tensor (bf16x3: both planes together)
static inline int hrn_esize(int dt) { return dt == HRN_BF16 ? 2 : 4; }

