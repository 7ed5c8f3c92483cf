/* Entry points of the measured-and-rejected experiments of rounds 2 and 3 (experiments/README.md).  They are NOT part of
 * libtgis_hip.so or of the drop-in boundary: experiments/build.py compiles the product sources (two of them through the
 * wrapper units experiments/csrc/{gptq,attention}_experiments.hip) plus the other units of experiments/csrc into
 * experiments/lib/libtgis_experiments.so, which exports everything below on top of
 * include/tgis_hip.h. */
#ifndef TGIS_EXPERIMENTS_H
#define TGIS_EXPERIMENTS_H
#include "../../include/tgis_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- add + RMSNorm as the FIRST PHASE of the GEMM behind it (decode, <= 32 rows, round 3) ----------------------------
 * `LlamaRMSNorm.forward` (flash_llama_modeling.py:132-152) followed by `gate_up_proj` + SiLU * up (:332-335), resp. by
 * `query_key_value` + rotary embedding + cache write (:251-268,282), as ONE launch: workgroup r normalises row r (the
 * arithmetic of tgis_rmsnorm_residual[_partial], bit for bit), all workgroups meet at a grid barrier while their weight
 * rings are already streaming, and the GEMM phase reads the normed rows with L1-bypassing loads.  `norm` describes the norm:
 * its input either as `x` or as the split-K `slabs` of the GEMM before (+ `slab_bias`), `residual` (may be NULL), `weight`,
 * `eps`, and its two outputs `y` (the normed activation [M, K]: also the GEMM's operand) and `res_out` (the residual stream).
 * Needs every workgroup of the launch resident at once: tgis_gptq_norm_gemm_ok says whether the shape qualifies on this
 * device (unsplit plan with M <= blocks <= CUs; not when TGIS_ALLOW_SHARED_GPU marks the GPU as shared between processes)
 * and allocates the library-owned barrier — call it once outside any stream capture.  All waits are bounded:
 * tgis_gptq_norm_gemm_status returns a give-up code (0 = none). */
typedef struct tgis_norm_in {
    const float* slabs;     /* [ceil(M/32)][num_slabs][32][slab_ld] fp32, or NULL */
    int num_slabs;
    int64_t slab_ld;
    const void* slab_bias;  /* bias of the GEMM that left the slabs, or NULL */
    const void* x;          /* f16 [M, K] when slabs == NULL */
    const void* residual;   /* f16 [M, K] or NULL */
    const void* weight;     /* f16 [K] */
    float eps;
    void* y;                /* f16 [M, K] out */
    void* res_out;          /* f16 [M, K] out */
} tgis_norm_in;
int tgis_gptq_norm_gemm_ok(int64_t M, int64_t K, int64_t N, int64_t groups, int act_order, int act);
int tgis_gptq_norm_gemm_status(int reset);
int tgis_gptq_norm_gate_up_f16(const tgis_norm_in* norm, const void* prepared, const void* bias, void* out, int64_t ldo,
                               int64_t M, int64_t K, int64_t N, int64_t groups, void* stream);
int tgis_gptq_norm_qkv_rope_f16(const tgis_norm_in* norm, const void* prepared, const void* bias, const int32_t* positions,
                                const int32_t* slots, const void* cos, const void* sin, void* q_out, int64_t ldq,
                                void* k_pool, void* v_pool, int64_t M, int64_t K, int64_t N, int64_t groups, int64_t H,
                                int64_t Hkv, int64_t D, void* stream);


/* ---- "lean" decode GEMM (round 3) -------------------------------------------------------------------------------
 * Same contract as tgis_gptq_gemm_f16 / tgis_gptq_gemm_f16_partial (the gemm_half_q_half call of
 * utils/gptq/exllamav2.py:139-144) for the shapes tgis_gptq_lean_ok() accepts — 1 <= M <= 32, group size 128, no
 * act-order, act 0 or 2 — on the SAME prepared image.  The nibbles go to the MFMA as 1024 + q / 64 + q (one VALU op per
 * pair instead of the 13-op dequantisation); the offsets and zero points are cancelled by one extra MFMA per 128-row
 * group whose operand is built from row sums of x, and the scale is applied once per group in fp32.  (q - z) * s is
 * therefore exact in fp32 here (exllamav2 rounds it to f16 once): results agree with tgis_gptq_gemm_f16 to that rounding.
 *
 * xs: fp32 [M][ldxs][2] — per row and 16 consecutive columns of x the pair {sum of x[k] over k % 4 < 2, sum over
 * k % 4 >= 2} — written by the producer of x: tgis_rmsnorm_residual*_xs, the act = 2 epilogue of this GEMM (xs_out, the
 * sums of its own [M, N/2] output for the down projection) or tgis_xsum_f16 for any other f16 matrix. */
/* 0, or the code a bounded in-kernel wait of the loader / consumer form left behind when it gave up (the launch then
 * finished with garbage results instead of hanging); `reset` clears it.  Synchronises with the device. */
int tgis_gptq_lean_status(int reset);
int tgis_gptq_lean_ok(int64_t M, int64_t K, int64_t N, int64_t groups, int act_order, int act);
int tgis_xsum_f16(const void* x, int64_t ldx, float* xs, int64_t ldxs, int64_t M, int64_t K, void* stream);
int tgis_gptq_gemm_f16_lean(const void* x, int64_t ldx, const float* xs, int64_t ldxs, const void* prepared,
                            const void* bias, void* out, int64_t ldo, float* xs_out, int64_t M, int64_t K, int64_t N,
                            int64_t groups, int act, void* workspace, int64_t workspace_bytes, void* stream);
int tgis_gptq_gemm_f16_partial_lean(const void* x, int64_t ldx, const float* xs, int64_t ldxs, const void* prepared,
                                    int64_t M, int64_t K, int64_t N, int64_t groups, float* slabs, int64_t slabs_bytes,
                                    int* num_slabs, int64_t* slab_ld, void* stream);


/* Decode step (one q token per sequence, cu_seqlens_q = 0..B) with the rotary embedding and the cache write of the new
 * token done in the attention launch's prologue: the work of tgis_rope_kv_write[_partial] followed by tgis_attn_paged,
 * with the same arithmetic and one launch less per layer (flash_llama_modeling.py:262-295 in one call).
 *   qkv [B, ld_qkv] (T): the UN-rotated output of the qkv projection, or — slabs != NULL — its split-K partial sums
 *   as left by tgis_*_gemm_partial (bias then = the projection's bias or NULL; qkv is ignored).
 *   cos / sin NULL: no rotation (GPT-BigCode), cache write only.  k_pool / v_pool are read AND written (slots[b]).
 * Every block computes the rotated q fragments it needs; the block that owns a sequence's last page writes the new
 * token's k and v into it before walking its pages.  The rotated q / k are not materialised anywhere else.
 * (Measured on MI355X: not faster than the two launches — DESIGN.md §6; the host mirror keeps them by default.) */
int tgis_attn_decode_rope(const void* qkv, int64_t ld_qkv, const float* slabs, int num_slabs, int64_t slab_ld,
                          const void* bias, const void* cos, const void* sin, const int32_t* positions,
                          const int32_t* slots, int rot_dim, void* k_pool, void* v_pool,
                          const int32_t* block_tables, int64_t max_pages, const int32_t* ctx_lens,
                          const int32_t* cu_seqlens_q, void* out, int64_t B, int H, int Hkv, int D,
                          int64_t max_ctx, float scale, int dtype, int num_splits, void* workspace,
                          int64_t workspace_bytes, void* stream);


/* ---- persistent decode tail of a Llama layer --------------------------------------------------------------------
 * Everything the decode step runs between two attention launches, in ONE launch (M <= 32 rows, one shard; either
 * int4 GPTQ linears in fp16, or dense f16 / bf16 linears):  o_proj -> add + RMSNorm -> gate_up (SiLU * up) -> down ->
 * add + RMSNorm [-> qkv of the NEXT layer -> rotary + KV-cache write of the next layer].  Dense layers go through the
 * same chain with tgis_dense_gemm(act = 2) / tgis_dense_gemm_partial.  Replaces, with bit-identical results, the call sequence
 * tgis_gptq_gemm_f16_partial / tgis_rmsnorm_residual_partial / tgis_gptq_gemm_f16(act=2) /
 * tgis_gptq_gemm_f16_partial / tgis_rmsnorm_residual_partial / tgis_gptq_gemm_f16_partial /
 * tgis_rope_kv_write_partial, i.e. the reference's FlashLlamaLayer tail + the next layer's head
 * (custom_modeling/flash_llama_modeling.py:285-297,383-385,332-335,368,251-282).  One workgroup per CU stays resident
 * for the whole launch; the phases are separated by grid barriers (8 group counters -> top counter -> 8 generation
 * words, relaxed agent-scope polling, every spin bounded) and hand their results over as sc1 write-through stores /
 * sc1 loads.  All buffers are caller-owned; slab buffers hold tgis_llama_decode_tail_slab_bytes(M, K, N, groups) bytes.
 * `qkv.prepared == NULL` ends the launch after the second norm (last layer: norm2_weight is then the final norm).
 * The GPU must not be shared with another process's persistent launch (tensor-parallel ranks on one device). */
typedef struct tgis_tail_linear {
    const void* prepared; /* image made by tgis_gptq_prepare or tgis_dense_prepare (gate_up: with flags bit 0) */
    const void* bias;     /* T [N] or NULL */
    int64_t K, N, groups; /* groups == 0: a dense f16 / bf16 image (all four linears of a layer are of one kind) */
} tgis_tail_linear;

typedef struct tgis_tail_args {
    int64_t M, hidden;
    float eps;
    const void* attn_out;    /* f16 [M, o_proj.K]: the attention output */
    const void* residual_in; /* f16 [M, hidden]: the residual stream before attention */
    tgis_tail_linear o_proj, gate_up, down, qkv;
    const void* norm1_weight; /* post-attention norm */
    const void* norm2_weight; /* next layer's input norm, or the final norm */
    void* y1;   /* f16 [M, hidden] scratch: normed input of gate_up */
    void* res1; /* f16 [M, hidden] scratch: residual after attention */
    void* act;  /* f16 [M, down.K] scratch: SiLU(gate) * up */
    void* y2;   /* f16 [M, hidden] out: normed hidden state (input of the next qkv / of the head) */
    void* res2; /* f16 [M, hidden] out: residual stream after the layer */
    float* slabs_o;
    float* slabs_down;
    float* slabs_qkv; /* NULL without qkv */
    void* qkv_out;    /* f16 [M, (H + 2 Hkv) D] out: rotated q, k, v of the next layer */
    const void* cos;  /* f16 [max_pos, rot_dim / 2] or NULL (no rotation) */
    const void* sin;
    const int32_t* positions; /* [M] */
    const int32_t* slots;     /* [M] physical cache slots of the new tokens */
    void* k_pool;             /* next layer's pools (see tgis_rope_kv_write) */
    void* v_pool;
    int H, Hkv, D, rot_dim;
    int dtype; /* TGIS_F16 / TGIS_BF16: element type T of every activation, norm weight, cos / sin table and KV pool
                * (int4 layers: TGIS_F16) */
} tgis_tail_args;

/* bytes of one slab buffer (slabs_o / slabs_down / slabs_qkv) for a linear of this shape; groups as in tgis_tail_linear */
int64_t tgis_llama_decode_tail_slab_bytes(int64_t M, int64_t K, int64_t N, int64_t groups);
/* 1 if the tail can run a layer of these shapes: only M, hidden and K / N / groups of the four linears are read
 * (qkv.K == 0 asks about a last layer).  The kernel exists for the plan signatures listed in csrc/decode_tail.hip. */
int tgis_llama_decode_tail_fits(const tgis_tail_args* shapes);
int tgis_llama_decode_tail(const tgis_tail_args* args, void* stream);
/* Debug aid: enable > 0 makes later launches record, per workgroup, 16 s_memrealtime stamps (100 MHz) at the edges of
 * the phases; `out` (may be NULL) receives [max_workgroups][16] stamps of the last launch; enable == 0 stops. */
int tgis_llama_decode_tail_trace(int enable, long long* out, int max_workgroups);
/* 1 if a grid barrier of this device ever hit its spin limit (its results are then invalid); `reset` re-arms. */
int tgis_llama_decode_tail_status(int reset);

#ifdef __cplusplus
}
#endif
#endif /* TGIS_EXPERIMENTS_H */
