// Shared by attention.hip (decode / short q) and attention_prefill.hip (long q): launch arguments and geometry.
#pragma once
#include "common.h"

struct AttnArgs {
    const void* q;
    int64_t ld_q;
    const void* kpool;
    const void* vpool;
    const int32_t* bt;
    int64_t max_pages;
    const int32_t* ctx_lens;
    const int32_t* cu_q;
    void* out;
    int out_frag;  // 1: out is [<= 32 tokens, H * D] in 32-row fragment order (xf_off in common.h), else row-major
    int H, Hkv, G, Gc, Gp, TQ, HC, NS;
    int Gp_shift;  // Gp is a power of two: column -> (q token, head of the group) by shift and mask, not by division
    int HCB;  // decode kernel: blocks per kv head along the 16-head chunks (HC / chunks per block)
    int xcd_remap;  // decode kernel, HCB > 1: the chunk blocks of a (sequence, split, kv head) group run on one XCD
    float scale_log2;
    float* ws_o;   // [total_q][H][NS][D]        (fused combine: [group][chunk][16 columns][NS][D])
    float* ws_ml;  // [total_q][H][NS][2]        (fused combine: [group][chunk][16 columns][NS][4], {m, l, -, -})
    unsigned* counters;  // fused combine: per-group arrival counters (zero between launches), or nullptr
    // Decode with the rotary embedding and the cache write of the new token in the prologue (tgis_attn_decode_rope):
    // q points at the un-rotated qkv activation [T, ld_q] (or is unused when qkv_slabs holds the qkv GEMM's split-K
    // partial sums); kpool / vpool are written at slots[t] before the block that owns the last page reads it.
    int fused_rope;            // 0: q is the rotated q of a finished qkv activation (the stand-alone rope kernel ran)
    const float* qkv_slabs;    // [S][32][qkv_slab_ld] or nullptr
    int qkv_S;
    int64_t qkv_slab_ld;
    const void* qkv_bias;      // T [(H + 2 Hkv) D] or nullptr (slab input only)
    const void* cosb;          // T [max_pos, rot / 2] or nullptr (no rotation: cache write only)
    const void* sinb;
    const int32_t* positions;  // [T]
    const int32_t* slots;      // [T]
    int rot;
};

constexpr float NEG_BIG = -1.0e30f;

// prefill kernel launcher (attention_prefill.hip); `a` carries the geometry filled in by tgis_attn_paged
int tgis_launch_attn_prefill(const AttnArgs& a, int64_t B, int Hkv, int D, int64_t max_q_len, int dtype, hipStream_t st);
