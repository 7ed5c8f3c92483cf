// KV page block layout for one (page, kv head), 32 tokens x D elements, and the "partial input" form of
// the qkv activation (fp32 split-K slabs of the qkv GEMM that the consumer sums itself).  Shared by rope_kv.hip (the
// stand-alone rotary + cache-write kernels) and attention.hip (the decode kernel that does both in its prologue).
//   K: [tile=tok>>4][D/8][16 tokens][8]      -> MFMA 16x16x32 A-fragments are 1 KiB contiguous loads
//   V: [4 column groups][D][8], token tok in column v_col(tok) = (i>>2)*8 + tile*4 + (i&3), i = tok&15 (column group =
//      v_col >> 3, slot = v_col & 7) -> a V^T A-fragment (16 rows d x 32 token columns) is four contiguous 256-byte runs of
//      whole cache lines, and the d run of ONE token (a decode step's write) is 16-byte strided: 16 cache lines per head
//      where the [D][32] order of rounds 1-3 touched 64 (round 4: the qkv + rotary launch 15.9 -> 12.6 us with cold pages)
#pragma once
#include "common.h"

__device__ __forceinline__ int64_t k_off(int tok, int d, int D) {
    return ((int64_t)(((tok >> 4) * (D >> 3) + (d >> 3)) * 16 + (tok & 15)) << 3) + (d & 7);
}
__device__ __forceinline__ int v_col(int tok) {
    int i = tok & 15;
    return (i >> 2) * 8 + (tok >> 4) * 4 + (i & 3);
}
// element offset of (token tok, dim d) inside the V block of one (page, kv head)
__device__ __forceinline__ int64_t v_off(int tok, int d, int D) {
    const int cp = v_col(tok);
    return ((int64_t)((cp >> 3) * D + d) << 3) + (cp & 7);
}

template <typename T> struct PartialIn {
    const float* slabs;  // [S][32][slab_ld] fp32 split-K partial sums of the qkv GEMM, or nullptr
    int S;
    int64_t slab_ld;
    const T* bias;
};

// 8 consecutive elements of row t starting at column col: from the model-dtype tensor (hp points at them), or the sum
// of the slabs (+ bias) rounded to the model dtype — bit-identical to reducing first and reading the tensor.
template <typename T>
__device__ __forceinline__ typename VecT<T>::x8 load_chunk(const T* hp, int64_t t, int col, const PartialIn<T>& pin) {
    using V8 = typename VecT<T>::x8;
    if (pin.slabs == nullptr) return ld16<V8>(hp);
    f32x4 lo, hi;
    sum_slabs8(pin.slabs + ((t >> 5) * pin.S * 32 + (t & 31)) * pin.slab_ld + col, 32 * pin.slab_ld, pin.S, lo, hi);
    V8 a;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float b0 = pin.bias ? to_f32(pin.bias[col + e]) : 0.f, b1 = pin.bias ? to_f32(pin.bias[col + 4 + e]) : 0.f;
        a[e] = from_f32<T>(lo[e] + b0);
        a[e + 4] = from_f32<T>(hi[e] + b1);
    }
    return a;
}
