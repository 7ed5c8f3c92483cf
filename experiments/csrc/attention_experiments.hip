// experiments/README.md: rotary embedding + cache write in the decode attention launch's prologue — built, verified bit-exact,
// measured slower than the two launches (cfg3 5.37 vs 4.92 ms/step); not part of libtgis_hip.so.
// This translation unit IS the product's csrc/attention.hip (included whole) plus the experimental entry point;
// experiments/build.py compiles it INSTEAD of the product file.
#include "../include/tgis_experiments.h"
#include "attention.hip"

extern "C" int tgis_attn_decode_rope(const void* qkv, int64_t ld_qkv, const float* slabs, int num_slabs, int64_t slab_ld,
                                     const void* bias, const void* cos, const void* sin, const int32_t* positions,
                                     const int32_t* slots, int rot_dim, void* k_pool, void* v_pool,
                                     const int32_t* block_tables, int64_t max_pages, const int32_t* ctx_lens,
                                     const int32_t* cu_seqlens_q, void* out, int64_t B, int H, int Hkv, int D,
                                     int64_t max_ctx, float scale, int dtype, int num_splits, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    TGIS_CHECK_ARG(slots, "tgis_attn_decode_rope: the cache write needs slots");
    TGIS_CHECK_ARG((cos == nullptr) == (sin == nullptr), "tgis_attn_decode_rope: cos and sin go together");
    TGIS_CHECK_ARG(!cos || (positions && rot_dim > 0 && rot_dim <= D && rot_dim % 16 == 0),
                   "tgis_attn_decode_rope: rot_dim must be a multiple of 16 and <= head_dim");
    TGIS_CHECK_ARG(slabs || (qkv && ld_qkv % 8 == 0 && ld_qkv >= (int64_t)(H + 2 * Hkv) * D),
                   "tgis_attn_decode_rope: needs the qkv activation or its split-K slabs");
    TGIS_CHECK_ARG(!slabs || (num_slabs >= 1 && slab_ld >= (int64_t)(H + 2 * Hkv) * D && slab_ld % 4 == 0),
                   "tgis_attn_decode_rope: needs a slab row stride >= (H + 2 Hkv) D");
    const FusedRope fr{slabs, num_slabs, slab_ld, slabs ? bias : nullptr, cos, sin, positions, slots, rot_dim};
    return attn_paged_impl(slabs ? nullptr : qkv, slabs ? 8 : ld_qkv, k_pool, v_pool, block_tables, max_pages, ctx_lens,
                           cu_seqlens_q, out, 0, B, H, Hkv, D, 1, max_ctx, scale, dtype, num_splits, workspace,
                           workspace_bytes, stream, &fr);
}
