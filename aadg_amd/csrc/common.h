// Shared host/device helpers for libaadg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>
#include <stdint.h>
#include "aadg_hip.h"

#define AADG_LAUNCH_CHECK()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

#define AADG_HIP_TRY(expr)                                   \
    do {                                                     \
        hipError_t e__ = (expr);                             \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

static inline size_t aadg_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// wave64 reductions (DPP/ds_swizzle lowered by the compiler from __shfl_xor)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// float32 -> bfloat16 bits, round to nearest even (NaN kept quiet); bfloat16 bits -> float32 is a 16-bit shift
// two float32 -> packed bfloat16 pair (lo in bits 0..15) with the gfx950 conversion instruction: round to nearest even,
// 16-byte streaming (non-temporal) stores for outputs that are written once and read again only after far more than a cache of
// other traffic: they do not displace the lines the kernel is about to (re-)read
typedef float aadg_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t aadg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void aadg_store_stream(float* p, float4 v) {
    const aadg_f32x4 q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<aadg_f32x4*>(p));
}
__device__ __forceinline__ void aadg_store_stream(void* p, uint4 v) {
    const aadg_u32x4 q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<aadg_u32x4*>(p));
}
// ... and 16-byte streaming loads for inputs that are read once
__device__ __forceinline__ float4 aadg_load_stream(const float* p) {
    const aadg_f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const aadg_f32x4*>(p));
    return make_float4(q.x, q.y, q.z, q.w);
}
__device__ __forceinline__ uint4 aadg_load_stream(const void* p) {
    const aadg_u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const aadg_u32x4*>(p));
    return make_uint4(q.x, q.y, q.z, q.w);
}
__device__ __forceinline__ void aadg_store_out(float* p, float4 v, bool stream) {
    if (stream) aadg_store_stream(p, v); else *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ void aadg_store_out(void* p, uint4 v, bool stream) {
    if (stream) aadg_store_stream(p, v); else *reinterpret_cast<uint4*>(p) = v;
}
// NaN kept quiet -- the same result as aadg_f2bf_bits() on each half, in one VALU instruction instead of ten
__device__ __forceinline__ uint32_t aadg_f2bf_pk(float lo, float hi) {
    typedef float aadg_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 aadg_bf16x2 __attribute__((ext_vector_type(2)));
    const aadg_f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, aadg_bf16x2));   // v_cvt_pk_bf16_f32
}
__device__ __forceinline__ uint32_t aadg_f2bf_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// float32 -> (hi, lo) bfloat16 halves: hi = bf16(x), lo = bf16(x - hi); x = hi + lo to ~2^-17 relative.  The "f32x3" convolution
// kernels multiply two such pairs as hi*hi + hi*lo + lo*hi on the bfloat16 matrix cores with float32 accumulation (the lo*lo term,
// ~2^-18 relative, is dropped): float32-grade products at a third of the bfloat16 MFMA rate = 5.3x the float32 MFMA rate of gfx950.
__device__ __forceinline__ void aadg_split4(float4 v, uint2& hi, uint2& lo) {
    const uint32_t h01 = aadg_f2bf_pk(v.x, v.y), h23 = aadg_f2bf_pk(v.z, v.w);
    const float r0 = v.x - __uint_as_float(h01 << 16), r1 = v.y - __uint_as_float(h01 & 0xFFFF0000u);
    const float r2 = v.z - __uint_as_float(h23 << 16), r3 = v.w - __uint_as_float(h23 & 0xFFFF0000u);
    hi = make_uint2(h01, h23);
    lo = make_uint2(aadg_f2bf_pk(r0, r1), aadg_f2bf_pk(r2, r3));
}
__device__ __forceinline__ void aadg_split1(float v, uint16_t& hi, uint16_t& lo) {
    const uint32_t h = aadg_f2bf_bits(v);
    hi = (uint16_t)h;
    lo = (uint16_t)aadg_f2bf_bits(v - __uint_as_float(h << 16));
}
