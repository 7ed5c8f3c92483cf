// RoPE (half-split / NeoX rotation) applied in place to the q and k heads of a fused qkv activation,
// fused with the scatter of k and v into the paged KV cache.
// Replaces rotary_emb.apply_rotary (utils/layers.py:466-472, called from
// custom_modeling/flash_llama_modeling.py:262-263) and the cache writes `layer_past[...] = kv`
// (flash_llama_modeling.py:268,282).
//
// KV page block layout for one (page, kv head), 32 tokens x D elements:
//   K: [tile=tok>>4][D/8][16 tokens][8]      -> MFMA 16x16x32 A-fragments are 1 KiB contiguous loads
//   V: [D][32], token tok at column (i>>2)*8 + tile*4 + (i&3), i = tok&15 -> V^T A-fragments likewise
#include "common.h"

namespace {

__device__ __forceinline__ int64_t k_off(int tok, int d, int D) {
    return ((int64_t)(((tok >> 4) * (D >> 3) + (d >> 3)) * 16 + (tok & 15)) << 3) + (d & 7);
}
__device__ __forceinline__ int v_col(int tok) {
    int i = tok & 15;
    return (i >> 2) * 8 + (tok >> 4) * 4 + (i & 3);
}

template <typename T>
__global__ __launch_bounds__(256) void rope_kv_kernel(T* qkv, int64_t ld, const T* __restrict__ cosb,
                                                      const T* __restrict__ sinb,
                                                      const int32_t* __restrict__ positions,
                                                      const int32_t* __restrict__ slots, T* __restrict__ kpool,
                                                      T* __restrict__ vpool, int H, int Hkv, int D, int rot) {
    using V8 = typename VecT<T>::x8;
    const int64_t t = blockIdx.x;
    T* row = qkv + t * ld;
    const int c8 = D >> 3;
    const int items = (H + 2 * Hkv) * c8;
    const int rh8 = rot >> 4;  // 8-element chunks in half the rotary span
    const int slot = slots ? slots[t] : 0;
    const int page = slot >> 5, tok = slot & 31;
    const T* cr = cosb ? cosb + (int64_t)positions[t] * (rot >> 1) : nullptr;
    const T* sr = cosb ? sinb + (int64_t)positions[t] * (rot >> 1) : nullptr;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int head = it / c8, j = it - head * c8;
        T* hp = row + head * D;
        const bool is_v = head >= H + Hkv;
        const bool is_k = head >= H && !is_v;
        const bool roped = cr != nullptr && !is_v;
        if (roped && j >= rh8 && j < 2 * rh8) continue;  // second half: handled with its partner
        V8 a = ld16<V8>(hp + j * 8);
        if (roped && j < rh8) {
            V8 b = ld16<V8>(hp + (j + rh8) * 8);
            V8 c = ld16<V8>(cr + j * 8), s = ld16<V8>(sr + j * 8);
            V8 o1, o2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x1 = to_f32(a[e]), x2 = to_f32(b[e]), cf = to_f32(c[e]), sf = to_f32(s[e]);
                o1[e] = from_f32<T>(x1 * cf - x2 * sf);
                o2[e] = from_f32<T>(x1 * sf + x2 * cf);
            }
            st16(hp + j * 8, o1);
            st16(hp + (j + rh8) * 8, o2);
            if (is_k && kpool) {
                T* kb = kpool + ((int64_t)page * Hkv + (head - H)) * 32 * D;
                st16(kb + k_off(tok, j * 8, D), o1);
                st16(kb + k_off(tok, (j + rh8) * 8, D), o2);
            }
        } else if (is_k && kpool) {
            T* kb = kpool + ((int64_t)page * Hkv + (head - H)) * 32 * D;
            st16(kb + k_off(tok, j * 8, D), a);
        } else if (is_v && vpool) {
            T* vb = vpool + ((int64_t)page * Hkv + (head - H - Hkv)) * 32 * D + v_col(tok);
#pragma unroll
            for (int e = 0; e < 8; ++e) vb[(int64_t)(j * 8 + e) * 32] = a[e];
        }
    }
}

}  // namespace

extern "C" int tgis_rope_kv_write(void* qkv, int64_t ld_qkv, const void* cos, const void* sin,
                                  const int32_t* positions, const int32_t* slots, void* k_pool, void* v_pool,
                                  int64_t T, int H, int Hkv, int D, int rot_dim, int dtype, void* stream) {
    TGIS_CHECK_ARG(qkv, "tgis_rope_kv_write: null qkv");
    TGIS_CHECK_ARG(H > 0 && Hkv >= 0 && D > 0 && D % 16 == 0, "tgis_rope_kv_write: head_dim must be a multiple of 16");
    TGIS_CHECK_ARG(ld_qkv % 8 == 0 && ld_qkv >= (int64_t)(H + 2 * Hkv) * D, "tgis_rope_kv_write: bad row stride");
    TGIS_CHECK_ARG((cos == nullptr) == (sin == nullptr), "tgis_rope_kv_write: cos and sin go together");
    TGIS_CHECK_ARG(!cos || (positions && rot_dim > 0 && rot_dim <= D && rot_dim % 16 == 0),
                   "tgis_rope_kv_write: rot_dim must be a multiple of 16 and <= head_dim");
    TGIS_CHECK_ARG((!k_pool && !v_pool) || slots, "tgis_rope_kv_write: cache write needs slots");
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_rope_kv_write: bad dtype");
    if (T == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    TgisTimedScope timed(TGIS_OP_ROPE_KV, st);
    if (dtype == TGIS_F16)
        hipLaunchKernelGGL(rope_kv_kernel<f16>, dim3((unsigned)T), dim3(256), 0, st, (f16*)qkv, ld_qkv,
                           (const f16*)cos, (const f16*)sin, positions, slots, (f16*)k_pool, (f16*)v_pool, H, Hkv,
                           D, rot_dim);
    else
        hipLaunchKernelGGL(rope_kv_kernel<bf16>, dim3((unsigned)T), dim3(256), 0, st, (bf16*)qkv, ld_qkv,
                           (const bf16*)cos, (const bf16*)sin, positions, slots, (bf16*)k_pool, (bf16*)v_pool, H,
                           Hkv, D, rot_dim);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}
