// Shared helpers for the gfx950 kernels of libtgis_hip.so (CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tgis_hip.h"

typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing (host) ---------------------------------------------------------------------
void tgis_set_error(const char* fmt, ...);
#define TGIS_CHECK_ARG(cond, ...)          \
    do {                                   \
        if (!(cond)) {                     \
            tgis_set_error(__VA_ARGS__);   \
            return TGIS_EINVAL;            \
        }                                  \
    } while (0)
#define TGIS_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            tgis_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                     \
            return TGIS_EHIP;                                                             \
        }                                                                                 \
    } while (0)
#define TGIS_CHECK_LAUNCH() TGIS_CHECK_HIP(hipGetLastError())

// Optional event timing around an op's launches (see tgis_timing_* in tgis_hip.h).
struct TgisTimedScope {
    int op;
    hipStream_t stream;
    void* slot;
    TgisTimedScope(int op, hipStream_t s);
    ~TgisTimedScope();
};

// ---- dtype traits (device) ---------------------------------------------------------------------
template <typename T> struct VecT;
template <> struct VecT<f16> { using x2 = f16x2; using x4 = f16x4; using x8 = f16x8; };
template <> struct VecT<bf16> { using x2 = bf16x2; using x4 = bf16x4; using x8 = bf16x8; };

__device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 16-byte global load/store of 8 two-byte elements.
template <typename V> __device__ __forceinline__ V ld16(const void* p) {
    return *reinterpret_cast<const V*>(p);
}
template <typename V> __device__ __forceinline__ void st16(void* p, V v) {
    *reinterpret_cast<V*>(p) = v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum of S split-K slabs for 8 consecutive columns: all loads of a bucket are issued before the first add (a
// runtime-trip loop would serialise one L2 round trip per slab).  stride = elements between slabs.
template <int SB>
__device__ __forceinline__ void sum_slabs_bucket(const float* sp, int64_t stride, int S, f32x4& lo, f32x4& hi) {
    f32x4 l[SB], h[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const float* q = sp + (int64_t)min(s, S - 1) * stride;
        l[s] = *reinterpret_cast<const f32x4*>(q);
        h[s] = *reinterpret_cast<const f32x4*>(q + 4);
    }
    lo = l[0];
    hi = h[0];
#pragma unroll
    for (int s = 1; s < SB; ++s) {
        if (s < S) {
            lo += l[s];
            hi += h[s];
        }
    }
    for (int s = SB; s < S; ++s) {
        lo += *reinterpret_cast<const f32x4*>(sp + (int64_t)s * stride);
        hi += *reinterpret_cast<const f32x4*>(sp + (int64_t)s * stride + 4);
    }
}
__device__ __forceinline__ void sum_slabs8(const float* sp, int64_t stride, int S, f32x4& lo, f32x4& hi) {
    if (S <= 1)
        sum_slabs_bucket<1>(sp, stride, S, lo, hi);
    else if (S <= 2)
        sum_slabs_bucket<2>(sp, stride, S, lo, hi);
    else if (S <= 4)
        sum_slabs_bucket<4>(sp, stride, S, lo, hi);
    else
        sum_slabs_bucket<8>(sp, stride, S, lo, hi);
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Element (row m, column k) of a [rows, K] activation stored in 32-row MFMA-FRAGMENT ORDER (TGIS_LD_FRAGMENTS in
// include/tgis_hip.h; K % 64 == 0): [row block m / 32][k64-step k / 64][i = k / 8 % 4][lane = 32 (k / 32 % 2) + m % 32][k % 8].
// One k64-step of a row block is four contiguous KiB, each the A operand of one v_mfma_f32_32x32x16 (lane l holds row l % 32,
// k-slots 8 (l / 32) ..+8) in the k order of the prepared weight images; 8 consecutive columns stay 16 contiguous bytes.
__host__ __device__ __forceinline__ int64_t xf_off(int64_t m, int64_t k, int64_t K) {
    return (m >> 5) * 32 * K + ((((k >> 6) << 2) + ((k >> 3) & 3)) * 64 + ((k >> 5) & 1) * 32 + (m & 31)) * 8 + (k & 7);
}

// GELU on an fp32 value (exact erf form, or the tanh approximation): ONE function for tgis_gelu and for the GEMM epilogues
// that apply it to their rounded output.  Not inlined: inside a caller the compiler contracts the expression (and the
// library's erf / tanh polynomials) with whatever surrounds it, and the two forms then differ in the last bit.
__device__ __attribute__((noinline)) inline float gelu_f32(float f, bool tanh_approx) {
    if (tanh_approx) {
        float inner = 0.7978845608028654f * (f + 0.044715f * f * f * f);
        return 0.5f * f * (1.f + tanhf(inner));
    }
    return 0.5f * f * (1.f + erff(f * 0.7071067811865476f));
}
