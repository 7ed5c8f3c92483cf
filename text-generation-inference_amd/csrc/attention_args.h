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
};

constexpr float NEG_BIG = -1.0e30f;

// prefill kernel launcher (attention_prefill.hip); `a` carries the geometry filled in by tgis_attn_paged
int tgis_launch_attn_prefill(const AttnArgs& a, int64_t B, int Hkv, int D, int64_t max_q_len, int dtype, hipStream_t st);
