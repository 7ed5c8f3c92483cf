// RoPE (half-split / NeoX rotation) applied in place to the q and k heads of a fused qkv activation,
// fused with the scatter of k and v into the paged KV cache.
// Replaces rotary_emb.apply_rotary (utils/layers.py:466-472, called from
// custom_modeling/flash_llama_modeling.py:262-263) and the cache writes `layer_past[...] = kv`
// (flash_llama_modeling.py:268,282).
//
// KV page block layout for one (page, kv head), 32 tokens x D elements:
//   K: [tile=tok>>4][D/8][16 tokens][8]      -> MFMA 16x16x32 A-fragments are 1 KiB contiguous loads
//   V: [4 column groups][D][8], token tok in column (i>>2)*8 + tile*4 + (i&3), i = tok&15 (kv_layout.h)
#include <algorithm>
#include "common.h"
#include "kv_layout.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void rope_kv_kernel(T* qkv, int64_t ld, const T* __restrict__ cosb,
                                                      const T* __restrict__ sinb,
                                                      const int32_t* __restrict__ positions,
                                                      const int32_t* __restrict__ slots, T* __restrict__ kpool,
                                                      T* __restrict__ vpool, int H, int Hkv, int D, int rot,
                                                      PartialIn<T> pin) {
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
    for (int it = blockIdx.y * blockDim.x + threadIdx.x; it < items; it += gridDim.y * blockDim.x) {
        const int head = it / c8, j = it - head * c8;
        T* hp = row + head * D;
        const bool is_v = head >= H + Hkv;
        const bool is_k = head >= H && !is_v;
        const bool roped = cr != nullptr && !is_v;
        if (roped && j >= rh8 && j < 2 * rh8) continue;  // second half: handled with its partner
        V8 a = load_chunk<T>(hp + j * 8, t, head * D + j * 8, pin);
        if (roped && j < rh8) {
            V8 b = load_chunk<T>(hp + (j + rh8) * 8, t, head * D + (j + rh8) * 8, pin);
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
        } else if (is_k) {
            if (pin.slabs) st16(hp + j * 8, a);
            if (kpool) {
                T* kb = kpool + ((int64_t)page * Hkv + (head - H)) * 32 * D;
                st16(kb + k_off(tok, j * 8, D), a);
            }
        } else if (is_v) {
            if (pin.slabs) st16(hp + j * 8, a);
            if (!vpool) continue;
            T* vb = vpool + ((int64_t)page * Hkv + (head - H - Hkv)) * 32 * D + v_off(tok, j * 8, D);
#pragma unroll
            for (int e = 0; e < 8; ++e) vb[e * 8] = a[e];
        } else if (pin.slabs) {
            st16(hp + j * 8, a);  // un-rotated q chunk (no rope / beyond the rotary span)
        }
    }
}

// Prefill form of the cache write: one block per (sequence, 32-token page, kv head).  The page's k rows are rotated and
// stored 16 tokens x 16 bytes at a time (contiguous 256-byte runs of the K layout); its v rows are transposed through
// LDS so that every store is a full 16-byte run of the [column group][D][8] layout (consecutive threads: consecutive runs).  The per-token kernel above issues 16-byte
// (k) and 2-byte (v) stores scattered over the page: fine for the 32 tokens of a decode step, ~4x slower than this on a
// 32k-token prefill.  Precondition: token i of sequence b sits at cache position i (a fresh prefill; its rotary
// position comes from `positions` like everywhere else).  Slots of the last page past the sequence end get zeros.
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_prefill_kernel(const T* __restrict__ qkv, int64_t ld,
                                                              const T* __restrict__ cosb, const T* __restrict__ sinb,
                                                              const int32_t* __restrict__ positions,
                                                              const int32_t* __restrict__ cu, const int32_t* __restrict__ bt,
                                                              int64_t max_pages, T* __restrict__ kpool,
                                                              T* __restrict__ vpool, int H, int Hkv, int D, int rot,
                                                              int pages_per_seq) {
    using V8 = typename VecT<T>::x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* vs = reinterpret_cast<T*>(smem);  // [32 tokens][D + 8]
    const int b = blockIdx.x / pages_per_seq, p = blockIdx.x % pages_per_seq, hk = blockIdx.y;
    const int t0 = cu[b], len = cu[b + 1] - t0;
    const int i0 = p * 32;
    if (i0 >= len) return;
    const int ntok = min(32, len - i0);
    const int page = bt[(int64_t)b * max_pages + p];
    const int tid = threadIdx.x;
    const int c8 = D >> 3, rh8 = rot >> 4;
    T* kb = kpool + ((int64_t)page * Hkv + hk) * 32 * D;
    T* vb = vpool + ((int64_t)page * Hkv + hk) * 32 * D;
    V8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (T)0.f;

    // ---- K: item = (token, 8-element chunk j); the rotary partner chunk j + rot/16 is handled with it ----------
    for (int it = tid; it < 32 * c8; it += 256) {
        const int tok = it & 31, j = it >> 5;
        const bool roped = cosb != nullptr;
        if (roped && j >= rh8 && j < 2 * rh8) continue;
        const bool pair = roped && j < rh8;
        V8 o1 = zero, o2 = zero;
        if (tok < ntok) {
            const int64_t t = t0 + i0 + tok;
            const T* kp = qkv + t * ld + (int64_t)(H + hk) * D;
            o1 = ld16<V8>(kp + j * 8);
            if (pair) {
                const V8 x2 = ld16<V8>(kp + (j + rh8) * 8);
                const T* cr = cosb + (int64_t)positions[t] * (rot >> 1);
                const T* sr = sinb + (int64_t)positions[t] * (rot >> 1);
                const V8 c = ld16<V8>(cr + j * 8), sn = ld16<V8>(sr + j * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a1 = to_f32(o1[e]), a2 = to_f32(x2[e]), cf = to_f32(c[e]), sf = to_f32(sn[e]);
                    o1[e] = from_f32<T>(a1 * cf - a2 * sf);
                    o2[e] = from_f32<T>(a1 * sf + a2 * cf);
                }
            }
        }
        st16(kb + k_off(tok, j * 8, D), o1);
        if (pair) st16(kb + k_off(tok, (j + rh8) * 8, D), o2);
    }

    // ---- V: stage [token][d] rows, store [column group][d][8 token columns] runs ----------------------------------------------
    const int rs = D + 8;
    for (int it = tid; it < 32 * c8; it += 256) {
        const int tok = it / c8, j = it - tok * c8;
        V8 v = zero;
        if (tok < ntok) v = ld16<V8>(qkv + (int64_t)(t0 + i0 + tok) * ld + (int64_t)(H + Hkv + hk) * D + j * 8);
        st16(vs + tok * rs + j * 8, v);
    }
    __syncthreads();
    for (int it = tid; it < D * 4; it += 256) {
        const int d = it % D, c = it / D;  // columns c*8 .. c*8+7 of row d = tokens {c*4+e, 16+c*4+e}, e < 4
        V8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = vs[(c * 4 + e) * rs + d];
            o[e + 4] = vs[(16 + c * 4 + e) * rs + d];
        }
        st16(vb + ((int64_t)c * D + d) * 8, o);
    }
}

}  // namespace

static int rope_launch(void* qkv, int64_t ld_qkv, const void* cos, const void* sin, const int32_t* positions,
                       const int32_t* slots, void* k_pool, void* v_pool, int64_t T, int H, int Hkv, int D, int rot_dim,
                       int dtype, void* stream, const float* slabs, int S, int64_t slab_ld, const void* bias) {
    TGIS_CHECK_ARG(qkv, "tgis_rope_kv_write: null qkv");
    TGIS_CHECK_ARG(H > 0 && Hkv >= 0 && D > 0 && D % 16 == 0, "tgis_rope_kv_write: head_dim must be a multiple of 16");
    TGIS_CHECK_ARG(ld_qkv % 8 == 0 && ld_qkv >= (int64_t)(H + 2 * Hkv) * D, "tgis_rope_kv_write: bad row stride");
    TGIS_CHECK_ARG((cos == nullptr) == (sin == nullptr), "tgis_rope_kv_write: cos and sin go together");
    TGIS_CHECK_ARG(!cos || (positions && rot_dim > 0 && rot_dim <= D && rot_dim % 16 == 0),
                   "tgis_rope_kv_write: rot_dim must be a multiple of 16 and <= head_dim");
    TGIS_CHECK_ARG((!k_pool && !v_pool) || slots, "tgis_rope_kv_write: cache write needs slots");
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_rope_kv_write: bad dtype");
    TGIS_CHECK_ARG(!slabs || (S >= 1 && slab_ld >= (int64_t)(H + 2 * Hkv) * D && slab_ld % 4 == 0),
                   "tgis_rope_kv_write_partial: needs a slab row stride >= (H + 2 Hkv) D");
    if (T == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    TgisTimedScope timed(TGIS_OP_ROPE_KV, st);
    // decode-sized T: spread one token's (H + 2 Hkv) * D/8 work items over several workgroups
    const int items = (H + 2 * Hkv) * (D >> 3);
    const unsigned gy = T <= 64 ? (unsigned)std::min(16, (items + 255) / 256) : 1u;
    const dim3 grid((unsigned)T, gy);
    if (dtype == TGIS_F16) {
        PartialIn<f16> pin{slabs, S, slab_ld, (const f16*)bias};
        hipLaunchKernelGGL(rope_kv_kernel<f16>, grid, dim3(256), 0, st, (f16*)qkv, ld_qkv,
                           (const f16*)cos, (const f16*)sin, positions, slots, (f16*)k_pool, (f16*)v_pool, H, Hkv,
                           D, rot_dim, pin);
    } else {
        PartialIn<bf16> pin{slabs, S, slab_ld, (const bf16*)bias};
        hipLaunchKernelGGL(rope_kv_kernel<bf16>, grid, dim3(256), 0, st, (bf16*)qkv, ld_qkv,
                           (const bf16*)cos, (const bf16*)sin, positions, slots, (bf16*)k_pool, (bf16*)v_pool, H,
                           Hkv, D, rot_dim, pin);
    }
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int tgis_rope_kv_write(void* qkv, int64_t ld_qkv, const void* cos, const void* sin,
                                  const int32_t* positions, const int32_t* slots, void* k_pool, void* v_pool,
                                  int64_t T, int H, int Hkv, int D, int rot_dim, int dtype, void* stream) {
    return rope_launch(qkv, ld_qkv, cos, sin, positions, slots, k_pool, v_pool, T, H, Hkv, D, rot_dim, dtype, stream,
                       nullptr, 0, 0, nullptr);
}

extern "C" int tgis_rope_kv_write_partial(const float* slabs, int num_slabs, int64_t slab_ld, const void* bias,
                                          void* qkv_out, int64_t ld_qkv, const void* cos, const void* sin,
                                          const int32_t* positions, const int32_t* slots, void* k_pool, void* v_pool,
                                          int64_t T, int H, int Hkv, int D, int rot_dim, int dtype, void* stream) {
    TGIS_CHECK_ARG(slabs, "tgis_rope_kv_write_partial: null slabs");
    return rope_launch(qkv_out, ld_qkv, cos, sin, positions, slots, k_pool, v_pool, T, H, Hkv, D, rot_dim, dtype,
                       stream, slabs, num_slabs, slab_ld, bias);
}

extern "C" int tgis_rope_kv_write_prefill(void* qkv, int64_t ld_qkv, const void* cos, const void* sin,
                                          const int32_t* positions, const int32_t* cu_seqlens,
                                          const int32_t* block_tables, int64_t max_pages, void* k_pool, void* v_pool,
                                          int64_t B, int64_t T, int64_t max_len, int H, int Hkv, int D, int rot_dim,
                                          int dtype, void* stream) {
    TGIS_CHECK_ARG(qkv && cu_seqlens && block_tables && k_pool && v_pool, "tgis_rope_kv_write_prefill: null tensor");
    TGIS_CHECK_ARG(B >= 0 && T >= 0 && max_len >= 0 && max_pages > 0, "tgis_rope_kv_write_prefill: bad sizes");
    TGIS_CHECK_ARG(H > 0 && Hkv > 0 && D > 0 && D % 16 == 0, "tgis_rope_kv_write_prefill: head_dim must be a multiple of 16");
    TGIS_CHECK_ARG((cos == nullptr) == (sin == nullptr), "tgis_rope_kv_write_prefill: cos and sin go together");
    TGIS_CHECK_ARG(!cos || (positions && rot_dim > 0 && rot_dim <= D && rot_dim % 16 == 0),
                   "tgis_rope_kv_write_prefill: rot_dim must be a multiple of 16 and <= head_dim");
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_rope_kv_write_prefill: bad dtype");
    if (B == 0 || T == 0) return TGIS_OK;
    // q heads: rotated in place by the per-token kernel (no cache traffic: Hkv = 0, no pools)
    int rc = rope_launch(qkv, ld_qkv, cos, sin, positions, nullptr, nullptr, nullptr, T, H, 0, D, rot_dim, dtype, stream,
                         nullptr, 0, 0, nullptr);
    if (rc != TGIS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int pps = (int)cdiv64(max_len, 32);
    TGIS_CHECK_ARG(pps <= max_pages && B * pps <= 2147483647LL && Hkv <= 65535, "tgis_rope_kv_write_prefill: grid too large");
    TgisTimedScope timed(TGIS_OP_ROPE_KV, st);
    const dim3 grid((unsigned)(B * pps), (unsigned)Hkv);
    const size_t lds = (size_t)32 * (D + 8) * 2;
    if (dtype == TGIS_F16)
        hipLaunchKernelGGL(rope_kv_prefill_kernel<f16>, grid, dim3(256), lds, st, (const f16*)qkv, ld_qkv, (const f16*)cos,
                           (const f16*)sin, positions, cu_seqlens, block_tables, max_pages, (f16*)k_pool, (f16*)v_pool, H,
                           Hkv, D, rot_dim, pps);
    else
        hipLaunchKernelGGL(rope_kv_prefill_kernel<bf16>, grid, dim3(256), lds, st, (const bf16*)qkv, ld_qkv,
                           (const bf16*)cos, (const bf16*)sin, positions, cu_seqlens, block_tables, max_pages,
                           (bf16*)k_pool, (bf16*)v_pool, H, Hkv, D, rot_dim, pps);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}
