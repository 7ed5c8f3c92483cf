/*
 * tgis_hip.h — C ABI of libtgis_hip.so, the MI355X (gfx950) native-kernel library that sits
 * where the reference's CUDA extension modules sit on the batched-decode hot path.
 *
 * Boundary: SURVEY.md §8(b) row #5.  The reference calls pybind/torch-extension functions with
 * at::Tensor arguments (flash_attn_2_cuda, dropout_layer_norm, rotary_emb, exllamav2_kernels);
 * this library exposes the same operations behind a plain C ABI: raw device pointers, int64
 * sizes/strides, float scalars and a hipStream_t passed as void*.  No torch types, no allocation
 * inside any call (callers pass workspaces; *_workspace_bytes queries say how large), no implicit
 * synchronisation.  Every function returns 0 on success or a negative TGIS_E* code;
 * tgis_last_error() returns a thread-local message for the last failure.
 *
 * dtype codes: TGIS_F16 (IEEE half) and TGIS_BF16.  All "T*" pointers below are 2-byte elements
 * of that dtype.  All pointers are device pointers unless the comment says host.
 *
 * Reference citations are relative to /root/reference/server/text_generation_server/.
 */
#ifndef TGIS_HIP_H
#define TGIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGIS_OK 0
#define TGIS_EINVAL (-1)   /* bad argument (shape, alignment, dtype) */
#define TGIS_EHIP (-2)     /* a HIP runtime call failed */
#define TGIS_EUNSUPPORTED (-3)
#define TGIS_ENOMEM (-4)

#define TGIS_F16 0
#define TGIS_BF16 1

/* KV page geometry: 32 tokens per page; per (page, k|v, kv_head) one
 * 32*head_dim element block in an MFMA-fragment-ready order. */
#define TGIS_KV_PAGE_TOKENS 32

/* ---- library info ------------------------------------------------------------------------- */
const char* tgis_version(void);          /* "tgis_hip x.y (gfx950)" */
const char* tgis_arch(void);             /* offload arch the kernels were compiled for */
const char* tgis_last_error(void);       /* thread-local, never NULL */
/* Forget the last failure, including the HIP runtime's sticky per-thread error: every launch in this library ends with
 * hipGetLastError(), so an error left behind by someone else's aborted stream capture would otherwise be reported by
 * the next, unrelated call. */
void tgis_clear_error(void);
int tgis_device_info(int device, int* num_cus, int64_t* hbm_bytes, char* name, int name_len);

/* Optional per-op device timing with HIP events recorded on the op's own stream.
 * tgis_timing_enable(1) makes every timed op (see TGIS_OP_*) bracket its kernel launch(es) with an
 * event pair; tgis_timing_read() synchronises those events and returns count and total
 * milliseconds since the last tgis_timing_reset().  Disabled by default (no events, no cost). */
#define TGIS_OP_GPTQ_GEMM 0
#define TGIS_OP_ATTN 1
#define TGIS_OP_DENSE_GEMM 2
#define TGIS_OP_NORM 3
#define TGIS_OP_ROPE_KV 4
#define TGIS_OP_ACT 5
#define TGIS_OP_SAMPLE 6
#define TGIS_OP_DECODE_TAIL 7
#define TGIS_OP_COUNT 8
int tgis_timing_enable(int on);
int tgis_timing_reset(void);
int tgis_timing_read(int op, int64_t* count, double* total_ms);

/* ---- GPTQ int4 linear (replaces exllamav2_kernels.make_q_matrix / gemm_half_q_half,
 *      utils/gptq/exllamav2.py:14-62,100-144; normative arithmetic utils/gptq/quant_linear.py:130-192) */

/* Bytes of the prepared (repacked) weight image for a [K,N] 4-bit matrix with `groups` groups. */
int64_t tgis_gptq_prepared_bytes(int64_t K, int64_t N, int64_t groups);

/* Repack GPTQ tensors into the kernel layout.  One-time, at load
 * (the reference does this in Ex4bitLinearV2.post_init, exllamav2.py:124-137).
 *   qweight [K/8, N] int32, qzeros [groups, N/8] int32, scales [groups, N] f16,
 *   g_idx [K] int32 or NULL (NULL = trivial k / groupsize), perm_out [K] int32 or NULL:
 *   when g_idx is not the trivial map (act-order) perm_out receives the row permutation that the
 *   activation must be gathered with (x'[:,k'] = x[:,perm[k']]); groupsize = K / groups.
 *   prepared: caller-owned buffer of tgis_gptq_prepared_bytes().  Requires K%32==0, N%32==0
 *   (same asserts as exllamav2.py:118-119) and (K/groups)%8==0.
 *   flags bit 0 (TGIS_GPTQ_GATE_UP): the matrix is a fused [gate | up] projection (N = 2 I); columns are
 *   interleaved in the image so that tgis_gptq_gemm_f16(act=2) can apply SiLU(gate)*up in its epilogue. */
#define TGIS_GPTQ_GATE_UP 1
int tgis_gptq_prepare(const int32_t* qweight, const int32_t* qzeros, const void* scales,
                      const int32_t* g_idx_host, int32_t* perm_out, int64_t K, int64_t N,
                      int64_t groups, int flags, void* prepared, void* stream);

/* Workspace for split-K partial sums + arrival counters of one gemm call. The counter region
 * (first 4096 bytes) must be zero before the first call; the kernel leaves it zero. */
/* Largest M for which tgis_gptq_gemm_f16 is a fused dequantise + MFMA kernel worth calling (above it the caller
 * dequantises once with tgis_gptq_dequant_f16 and uses a library GEMM, exllamav2.py:87 "M > 50"): M <= 64 streams the
 * weights once per pass, 64 < M <= 256 in 64-row passes, and — for K % 64 == 0, group size 64 * 2^n, no act-order, act != 1 —
 * a tall kernel (one dequantisation per 128 rows, 700-900 TFLOP/s, no scratch copy of W) takes over up to a few thousand
 * rows. */
int64_t tgis_gptq_gemm_fused_rows(int64_t K, int64_t groups, int act_order, int act);
int64_t tgis_gptq_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N);

/* Activation layout between the decode step's own kernels (round 4).  A leading dimension of TGIS_LD_FRAGMENTS says that a
 * [M <= 64, K] f16 activation (K % 64 == 0; the buffer holds ceil(M / 32) * 32 * K elements) is stored in the order the MFMA
 * reads it — [row block m/32][k64-step][i = k/8 % 4][lane = 32 (k/32 % 2) + m % 32][k % 8], i.e. element (m, k) at
 * (m/32) * 32 K + ((k/64 * 4 + k/8 % 4) * 64 + 32 (k/32 % 2) + m % 32) * 8 + k % 8 — instead of row-major.  Producers that can write it:
 * tgis_rmsnorm_residual[_partial] (ldy), tgis_attn_paged (ld_out), tgis_gptq_gemm_f16 with act = 2 (ldo).  Consumers:
 * tgis_gptq_gemm_f16, tgis_gptq_gemm_f16_partial, tgis_gptq_gemm_rope_f16 (ldx), which then run the kernel of
 * csrc/gptq_wide_body.h: each k64-step of the activation is four contiguous KiB straight into the A operand, no LDS staging
 * (7B shapes, 32 rows: qkv + rope 16.9 -> 12 us, o 6.0 -> 5.3, gate_up 16.8 -> 14.4, down 10.5 -> 9.2).  Same arithmetic as
 * the row-major path ((q - z) * s rounded to f16 once, fp32 accumulation); only the summation order over k differs.
 * tgis_gptq_fragments_ok: 1 if the GEMM (act 0, 2, or 3 = the rope epilogue) takes such an activation and is expected to be
 * faster with it (1 <= M <= 64, no act-order, groups of 64 * 2^n rows; act 2 / 3 additionally >= 128 workgroups). */
#define TGIS_LD_FRAGMENTS ((int64_t)-32)
int tgis_gptq_fragments_ok(int64_t M, int64_t K, int64_t N, int64_t groups, int act_order, int act);

/* out[M,N] f16 = x[M,K] f16 @ dequant(W)[K,N] (+ bias[N] f16 if non-NULL); fp32 accumulate.
 * Fused int4-dequant MFMA kernel for any M (rows are processed in slabs of 32).
 * x row stride ldx, out row stride ldo (elements).  perm (int32 [K] or NULL) gathers x columns
 * for act-order matrices.  act: 0 = none; 1 = x is [M,2K] and the kernel consumes silu(x[:, :K]) * x[:, K:]
 * while staging it; 2 = the image was prepared with TGIS_GPTQ_GATE_UP and out is [M, N/2] =
 * silu(gate) * up applied in the epilogue (both fuse LlamaMLP's activation, flash_llama_modeling.py:332-335). */
int tgis_gptq_gemm_f16(const void* x, int64_t ldx, const void* prepared, const void* bias,
                       const int32_t* perm, void* out, int64_t ldo, int64_t M, int64_t K, int64_t N,
                       int64_t groups, int act, void* workspace, int64_t workspace_bytes, void* stream);

/* Deferred-reduce variant: instead of the f16 result, the S split-K partial sums are left in
 * `slabs` as fp32 [ceil(M/32)][S][32][ld] (ld = N rounded up to 32, returned in *slab_ld; S in *num_slabs, >= 1) for a
 * consumer that sums them in its prologue (tgis_rmsnorm_residual_partial / tgis_rope_kv_write_partial), which
 * removes the reduce launch.  The consumer rounds the sum (+bias) to f16 first, so results are bit-identical to
 * tgis_gptq_gemm_f16 followed by the plain consumer.  slabs must hold tgis_gptq_gemm_partial_bytes(M,K,N) bytes. */
int64_t tgis_gptq_gemm_partial_bytes(int64_t M, int64_t K, int64_t N);
int tgis_gptq_gemm_f16_partial(const void* x, int64_t ldx, const void* prepared, const int32_t* perm, int64_t M,
                               int64_t K, int64_t N, int64_t groups, int act, float* slabs, int64_t slabs_bytes,
                               int* num_slabs, int64_t* slab_ld, void* stream);

/* ---- fused qkv projection + rotary embedding + cache write (decode, round 3) ------------------------------------------
 * One launch for `qkv = query_key_value(x)`, `rotary_emb(q, k, cos, sin)` and `layer_past[slot] = (k, v)`
 * (flash_llama_modeling.py:251-268,282) at decode sizes: the GEMM keeps the whole k range in a block (no split-K slabs)
 * and its epilogue rounds the sum (+ bias) to f16, rotates q / k heads in fp32 (the arithmetic of tgis_rope_kv_write) and
 * stores q to q_out[M, ldq] and k / v into the cache pages of slots[m] (page layouts of tgis_rope_kv_write).
 * `prepared` is a SECOND image of the qkv weight made with flags = TGIS_GPTQ_ROPE_IMAGE(D, H + Hkv): inside each rotated
 * head a 32-column tile holds 16 dims and their 16 rotation partners, so every wave owns complete rotation pairs.
 * Full rotary span only (rot_dim == D).  tgis_gptq_rope_ok: whether the launch exists for the shape (1 <= M <= 64, groups
 * of 64 * 2^n rows, no act-order, D % 32 == 0) AND is expected to beat the two launches it replaces (its unsplit plan
 * still covers the chip: >= 128 workgroups); the entry point itself only checks the former. */
#define TGIS_GPTQ_ROPE_IMAGE(D, rotated_heads) (2 | ((int)(D) << 8) | ((int)(rotated_heads) << 20))
int tgis_gptq_rope_ok(int64_t M, int64_t K, int64_t N, int64_t groups, int act_order, int64_t D);
int tgis_gptq_gemm_rope_f16(const void* x, int64_t ldx, const void* prepared, const void* bias, const int32_t* positions,
                            const int32_t* slots, const void* cos, const void* sin, void* q_out, int64_t ldq,
                            void* k_pool, void* v_pool, int64_t M, int64_t K, int64_t N, int64_t groups, int64_t H,
                            int64_t Hkv, int64_t D, void* stream);

/* The same launch for dense (f16 / bf16) qkv weights: `prepared` from tgis_dense_prepare with
 * flags = TGIS_GPTQ_ROPE_IMAGE(D, H + Hkv); cos / sin in the model dtype; 1 <= M <= 64. */
int tgis_dense_rope_ok(int64_t M, int64_t K, int64_t N, int64_t D);
int tgis_dense_gemm_rope(const void* x, int64_t ldx, const void* prepared, const void* bias, const int32_t* positions,
                         const int32_t* slots, const void* cos, const void* sin, void* q_out, int64_t ldq, void* k_pool,
                         void* v_pool, int64_t M, int64_t K, int64_t N, int64_t H, int64_t Hkv, int64_t D, int dtype,
                         void* stream);

/* Full dequantisation to a dense f16 [K,N] matrix (row-major), the "temp_dq" path the reference
 * uses for M > 50 before a library GEMM (exllamav2.py:65-66,87). */
int tgis_gptq_dequant_f16(const void* prepared, void* w_out, int64_t K, int64_t N, int64_t groups,
                          int flags, void* stream);

/* ---- dense skinny GEMM (replaces F.linear / torch.mm at decode sizes, utils/layers.py:110-111,
 *      lm_head utils/layers.py:261) -------------------------------------------------------------- */
int64_t tgis_dense_prepared_bytes(int64_t N, int64_t K);
/* Repack a torch-Linear weight W[N,K] (row-major, dtype) into 32-column MFMA tiles.
 * flags bit 0: W is the Llama MLP's [gate rows | up rows] (N = 2 I, I a multiple of 16): tile t then holds gate rows
 * 16 t .. 16 t + 15 and the matching up rows, for the act = 2 epilogue of tgis_dense_gemm. */
int tgis_dense_prepare(const void* w, int64_t N, int64_t K, int dtype, int flags, void* prepared, void* stream);
int64_t tgis_dense_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N);
/* out[M,N] = x[M,K] @ W^T (+bias).  out_f32 != 0 writes float32 output (logits).
 * act 0: plain.  act 1: x is [M, 2K] (gate | up) and the operand is silu(gate) * up, formed while it is staged
 * (down_proj after an un-fused gate_up).  act 2: the image has flags bit 0 and out is [M, N/2] =
 * silu(x @ Wgate^T) * (x @ Wup^T), each factor rounded to the model dtype as the reference's eager ops do
 * (flash_llama_modeling.py:332-335); the down projection then runs with act 0.
 * act 4 / 5: out = gelu(T(x @ W^T + bias)) in the model dtype, exact erf form / tanh approximation — the `self.act(...)`
 * behind `c_fc` (flash_santacoder_modeling.py:284-306) applied to the rounded output where it is finished (the epilogue of an
 * unsplit plan, the split-K reduce otherwise): bit-identical to tgis_dense_gemm(act 0) followed by tgis_gelu. */
int tgis_dense_gemm(const void* x, int64_t ldx, const void* prepared, const void* bias, void* out,
                    int64_t ldo, int64_t M, int64_t K, int64_t N, int dtype, int out_f32, int act,
                    void* workspace, int64_t workspace_bytes, void* stream);
/* Deferred split-K, as tgis_gptq_gemm_f16_partial: the fp32 partial sums [ceil(M/32)][num_slabs][32][slab_ld] are
 * left for the consumer kernel (tgis_rmsnorm_residual_partial / tgis_rope_kv_write_partial), which adds the bias. */
int64_t tgis_dense_gemm_partial_bytes(int64_t M, int64_t K, int64_t N);
int tgis_dense_gemm_partial(const void* x, int64_t ldx, const void* prepared, int64_t M, int64_t K, int64_t N,
                            int dtype, int act, float* slabs, int64_t slabs_bytes, int* num_slabs,
                            int64_t* slab_ld, void* stream);

/* ---- fused residual-add + RMSNorm / LayerNorm (replaces dropout_layer_norm.dropout_add_ln_fwd,
 *      custom_modeling/flash_llama_modeling.py:132-152, utils/layers.py:376-396) ---------------- */
/* res_out = x (+ residual); y = res_out * rsqrt(mean(res_out^2) + eps) * weight.
 * residual may be NULL (first layer).  res_out may alias residual or x.  fp32 statistics.
 * ldy: 0 or hidden (y row-major), or TGIS_LD_FRAGMENTS (rows <= 64, hidden % 64 == 0: y feeds an int4 GEMM of the decode
 * step, see TGIS_LD_FRAGMENTS above). */
int tgis_rmsnorm_residual(const void* x, const void* residual, const void* weight, void* y, int64_t ldy,
                          void* res_out, int64_t rows, int64_t hidden, float eps, int dtype,
                          void* stream);
/* Same as tgis_rmsnorm_residual with x given as split-K partial sums: x = f16(sum_s slabs[s][row][:] (+ bias)).
 * rows <= 32. */
int tgis_rmsnorm_residual_partial(const float* slabs, int num_slabs, int64_t slab_ld, const void* bias,
                                  const void* residual, const void* weight, void* y, int64_t ldy, void* res_out,
                                  int64_t rows, int64_t hidden, float eps, int dtype, void* stream);
int tgis_layernorm_residual(const void* x, const void* residual, const void* weight, const void* bias,
                            void* y, void* res_out, int64_t rows, int64_t hidden, float eps,
                            int dtype, void* stream);
/* tgis_layernorm_residual with x given as split-K partial sums (see tgis_rmsnorm_residual_partial): x = model-dtype
 * rounding of sum_s slabs[s][row][:] (+ xbias, the producing linear's bias).  Removes the reduce launch after the
 * c_proj linears of flash_santacoder_modeling.py:255-307. */
int tgis_layernorm_residual_partial(const float* slabs, int num_slabs, int64_t slab_ld, const void* xbias,
                                    const void* residual, const void* weight, const void* bias, void* y,
                                    void* res_out, int64_t rows, int64_t hidden, float eps, int dtype,
                                    void* stream);

/* ---- RoPE + KV-cache write (replaces rotary_emb.apply_rotary + the index_put at
 *      flash_llama_modeling.py:262-268,282; utils/layers.py:466-472) ---------------------------- */
/* qkv [T, (H + 2*Hkv)*D] (row stride ld_qkv): rotates q heads and k heads in place with the
 * half-split (NeoX) rotation using cos/sin tables [max_pos, rot_dim/2] of the model dtype gathered by
 * positions[T] (int32), then writes k and v of token t into KV page slot slots[t]
 * (= page_id*32 + offset).  cos == NULL skips the rotation (learned-position models).
 * k_pool / v_pool: this layer's K and V page pools, [num_pages][Hkv][32*D] each.  Inside a (page, kv head) block of
 * 32 tokens x D: K as [token >> 4][D / 8][16 tokens][8], V as [4 column groups][D][8] with token t in column
 * (i >> 2) * 8 + (t >> 4) * 4 + (i & 3), i = t & 15 (csrc/kv_layout.h, DESIGN.md section 3: the orders in which the
 * attention kernels' MFMA fragments are whole cache lines).  The pools are opaque to callers; oracle/ops_ref.py's
 * kv_page_unpack reads them back for tests. */
int tgis_rope_kv_write(void* qkv, int64_t ld_qkv, const void* cos, const void* sin,
                       const int32_t* positions, const int32_t* slots, void* k_pool, void* v_pool,
                       int64_t T, int H, int Hkv, int D, int rot_dim, int dtype, void* stream);

/* Same with the qkv activation given as split-K partial sums: qkv_out[T, ld_qkv] receives
 * f16(sum_s slabs[s] (+ bias)) with q,k rotated; k,v go to the cache. */
int tgis_rope_kv_write_partial(const float* slabs, int num_slabs, int64_t slab_ld, const void* bias, void* qkv_out,
                               int64_t ld_qkv, const void* cos, const void* sin, const int32_t* positions,
                               const int32_t* slots, void* k_pool, void* v_pool, int64_t T, int H, int Hkv, int D,
                               int rot_dim, int dtype, void* stream);

/* Prefill form of tgis_rope_kv_write for fresh sequences (token i of sequence b is cache position i): q is rotated in
 * place; k (rotated) and v go to the cache page by page — full 16-byte runs of the page layouts instead of the
 * per-token scatter — with the slots of a last partial page zeroed.  k/v inside `qkv` are left as they came.
 *   cu_seqlens [B+1]: token offsets; block_tables [B, max_pages]; T = total tokens; max_len = longest sequence. */
int tgis_rope_kv_write_prefill(void* qkv, int64_t ld_qkv, const void* cos, const void* sin, const int32_t* positions,
                               const int32_t* cu_seqlens, const int32_t* block_tables, int64_t max_pages, void* k_pool,
                               void* v_pool, int64_t B, int64_t T, int64_t max_len, int H, int Hkv, int D, int rot_dim,
                               int dtype, void* stream);

/* ---- paged attention, prefill and decode (replaces flash_attn_2_cuda.varlen_fwd,
 *      utils/flash_attn.py:43-78) ---------------------------------------------------------------- */
/* Number of key-range splits the launcher will use for this shape (so callers can size workspace). */
int tgis_attn_num_splits(int64_t B, int Hkv, int H, int64_t max_q_len, int64_t max_ctx);
int64_t tgis_attn_workspace_bytes(int64_t total_q_tokens, int H, int Hkv, int D, int num_splits);
/* Causal softmax(q k^T * scale) v over the paged cache.
 *   q: [total_q, H, D] with token stride ld_q (elements); out: [total_q, H*D] contiguous (ld_out = 0 or
 *   H*D), or — decode, B <= 64 — in fragment order for the o_proj GEMM (ld_out = TGIS_LD_FRAGMENTS).
 *   cu_seqlens_q [B+1] int32: q token offsets per sequence (decode: arange).
 *   ctx_lens [B] int32: tokens of each sequence present in the cache INCLUDING the q tokens.
 *   block_tables [B, max_pages] int32: page ids.  q token i of sequence b sits at position
 *   ctx_lens[b] - q_len_b + i and attends to cache positions <= its own.
 *   max_q_len / max_ctx are launch-shape bounds (host ints). */
int tgis_attn_paged(const void* q, int64_t ld_q, const void* k_pool, const void* v_pool,
                    const int32_t* block_tables, int64_t max_pages, const int32_t* ctx_lens,
                    const int32_t* cu_seqlens_q, void* out, int64_t ld_out, int64_t B, int H, int Hkv, int D,
                    int64_t max_q_len, int64_t max_ctx, float scale, int dtype, int num_splits,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---- elementwise --------------------------------------------------------------------------------- */
/* out[T,I] = act(gate_up[T,0:I]) * gate_up[T,I:2I]; act 1 = SiLU (flash_llama_modeling.py:332-335). */
int tgis_act_mul(const void* gate_up, void* out, int64_t T, int64_t I, int act, int dtype, void* stream);
/* out[T,I] = gelu(x) ; approx 1 = tanh (flash_santacoder_modeling.py:303-307). */
int tgis_gelu(const void* x, void* out, int64_t n, int tanh_approx, int dtype, void* stream);
/* out[T,E] = table[ids[T]] (+ pos_table[positions[T]] if non-NULL); ids int64. Out-of-range id -> 0 row
 * (TensorParallelEmbedding's null row, utils/layers.py:325-357); ids are offset by -id_offset first. */
int tgis_embedding(const int64_t* ids, const void* table, const int32_t* positions,
                   const void* pos_table, void* out, int64_t T, int64_t E, int64_t vocab_rows,
                   int64_t id_offset, int dtype, void* stream);
/* Decode-step bookkeeping in one launch (flash_causal_lm.py:457-458,499 on device):
 * positions[b] (int32) -> slots[b] = block_tables[b][pos/32]*32 + pos%32 ; ctx_lens[b] = pos+1. */
int tgis_decode_slots(const int32_t* positions, const int32_t* block_tables, int64_t max_pages,
                      int32_t* slots, int32_t* ctx_lens, int64_t B, void* stream);
/* What follows a decode step on the device, in one launch (flash_causal_lm.py:457 `cu_seqlens.add_(cu_seqlens_q)`, :499
 * `position_ids += 1`, :533-535 `all_input_ids_tensor.scatter_(1, position_ids, next ids)`): position_ids int64 [B] += 1;
 * all_input_ids[b][position_ids[b]] = ids[b] (row stride ld_all, NULL: skip); ids_copy [B] = ids (NULL: skip; the caller's
 * next `input_ids`, since `ids` is a buffer the next step overwrites); cu_seqlens int32 [B + 1] += cu_seqlens_q (NULL:
 * skip); stage_ids int64 [B] / stage_positions int32 [B]: the same new ids / positions once more, e.g. into the static
 * input buffers of a captured decode step (NULL: skip). */
int tgis_decode_advance(const int64_t* ids, int64_t* ids_copy, int64_t* position_ids, int64_t* all_input_ids,
                        int64_t ld_all, int32_t* cu_seqlens, const int32_t* cu_seqlens_q, int64_t* stage_ids,
                        int32_t* stage_positions, int64_t B, void* stream);

/* ---- greedy sampling (Greedy + log_softmax + gather, utils/tokens.py:44-46,238-271,388-397) ------ */
/* Per row: token = argmax (lowest id on ties), logprob = logit[token] - logsumexp(row).
 * logits [B,V] f32 (logits_f32 != 0) or model dtype. ids_out int64 [B], logprob_out f32 [B].
 * scratch (may be NULL): tgis_argmax_scratch_bytes(B) bytes of device memory the call may use until it completes; with it
 * a small batch's rows are split over several workgroups each (two launches; same ids, logprobs within summation order). */
int64_t tgis_argmax_scratch_bytes(int64_t B);
int tgis_argmax_logprob(const void* logits, int64_t ld, int64_t B, int64_t V, int logits_f32, int dtype,
                        int64_t* ids_out, float* logprob_out, void* scratch, int64_t scratch_bytes, void* stream);

/* ---- next-token chooser for a heterogeneous batch -------------------------------------------------
 * One launch for what HeterogeneousNextTokenChooser.__call__ (utils/tokens.py:242-270) does with the
 * Heterogeneous* processors (utils/logits_process.py:93-402) and Sampling / HeterogeneousSampling
 * (utils/tokens.py:32-41,336-385), in the reference's order: EOS mask or length penalty, repetition
 * penalty, temperature, top-k, top-p, typical-p, then argmax (greedy rows) or a categorical draw.
 *   logits  [B,V] f32 (read only);  scores [B,V] f32 out: the warped scores, -inf where filtered — the
 *           tensor get_token_info (utils/tokens.py:388-425) takes top-n tokens and ranks from.
 *   Per-row parameter arrays, each may be NULL (= processor not instantiated): temperature; top_k
 *   (0 = off); top_p_cut = 1 - top_p as the host rounded it (<= 0 = off); typical_p (>= 1 = off);
 *   rep_penalty (1 = off) with input_ids [B,L] int64 (row stride ld_ids; padding included, as the
 *   reference passes it) and exclude_id (the id the penalty leaves alone, -1 = none);
 *   eos_adjust [B,2] = (mode, factor): mode 1 sets scores[eos_id] = -inf (min_new_tokens), mode 2
 *   scores[eos_id] += |scores[eos_id]| * factor (length penalty); do_sample [B] (NULL = all greedy).
 *   rng [B,2] uint64 (seed, offset): the request's Philox4x32-10 stream; offset += 1 per draw.
 * Outputs: next_ids int64 [B]; next_logprob f32 [B] = log_softmax(scores)[id]; lse f32 [B] =
 * logsumexp(scores) (log_softmax of a row = scores - lse). */
int tgis_warp_sample(const float* logits, int64_t ld_logits, float* scores, int64_t ld_scores, int64_t B,
                     int64_t V, const float* temperature, const int* top_k, const float* top_p_cut,
                     const float* typical_p, const float* rep_penalty, const int64_t* input_ids,
                     int64_t ld_ids, int64_t L, int64_t exclude_id, const float* eos_adjust, int64_t eos_id,
                     const int* do_sample, uint64_t* rng, int64_t* next_ids, float* next_logprob, float* lse,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TGIS_HIP_H */
