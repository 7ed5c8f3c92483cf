// Small bandwidth-bound helpers of the decode step: SiLU*mul / GELU, embedding gather, decode slot
// bookkeeping and greedy argmax + logprob.  All use 16-byte vector accesses.
#include "common.h"

namespace {

template <typename T>
__global__ void act_mul_kernel(const T* __restrict__ gu, T* __restrict__ out, int64_t T_, int64_t I) {
    using V8 = typename VecT<T>::x8;
    const int64_t c8n = I >> 3;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T_ * c8n) return;
    int64_t t = idx / c8n, c = idx - t * c8n;
    V8 g = ld16<V8>(gu + t * 2 * I + c * 8);
    V8 u = ld16<V8>(gu + t * 2 * I + I + c * 8);
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float gf = to_f32(g[e]);
        float sl = gf / (1.f + __expf(-gf));
        // eager torch: act() result rounded to the model dtype, then the product rounded again
        o[e] = from_f32<T>(to_f32(from_f32<T>(sl)) * to_f32(u[e]));
    }
    st16(out + t * I + c * 8, o);
}

template <typename T>
__global__ void gelu_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t n, int tanh_approx) {
    using V8 = typename VecT<T>::x8;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx * 8 >= n) return;
    V8 v = ld16<V8>(x + idx * 8);
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] = from_f32<T>(gelu_f32(to_f32(v[e]), tanh_approx != 0));
    }
    st16(out + idx * 8, o);
}

template <typename T>
__global__ void embedding_kernel(const int64_t* __restrict__ ids, const T* __restrict__ table,
                                 const int32_t* __restrict__ positions, const T* __restrict__ pos_table,
                                 T* __restrict__ out, int64_t E, int64_t vocab_rows, int64_t id_offset) {
    using V8 = typename VecT<T>::x8;
    const int64_t t = blockIdx.x;
    const int64_t id = ids[t] - id_offset;
    const bool valid = id >= 0 && id < vocab_rows;
    const int64_t c8n = E >> 3;
    for (int64_t c = threadIdx.x; c < c8n; c += blockDim.x) {
        V8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (T)0.f;
        if (valid) v = ld16<V8>(table + id * E + c * 8);
        if (pos_table) {
            V8 p = ld16<V8>(pos_table + (int64_t)positions[t] * E + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(p[e]));
        }
        st16(out + t * E + c * 8, v);
    }
}

__global__ void decode_slots_kernel(const int32_t* __restrict__ positions, const int32_t* __restrict__ bt,
                                    int64_t max_pages, int32_t* __restrict__ slots, int32_t* __restrict__ ctx,
                                    int64_t B) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int pos = positions[b];
    int page = bt[b * max_pages + (pos >> 5)];
    slots[b] = page * TGIS_KV_PAGE_TOKENS + (pos & 31);
    ctx[b] = pos + 1;
}

// The device side of what follows a decode step (reference flash_causal_lm.py:457,499,533-535), one launch:
// position_ids += 1; all_input_ids[b][position] = new id; a private copy of the ids (the chooser's / graph's buffer is
// reused by the next step); cu_seqlens += cu_seqlens_q; optionally the next step's inputs straight into a decode graph's
// static buffers.
__global__ void decode_advance_kernel(const int64_t* __restrict__ ids, int64_t* __restrict__ ids_copy,
                                      int64_t* __restrict__ position_ids, int64_t* __restrict__ all_ids, int64_t ld_all,
                                      int32_t* __restrict__ cu_seqlens, const int32_t* __restrict__ cu_q,
                                      int64_t* __restrict__ stage_ids, int32_t* __restrict__ stage_pos, int64_t B) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > B) return;
    if (cu_seqlens) cu_seqlens[b] += cu_q[b];  // B + 1 entries
    if (b == B) return;
    const int64_t id = ids[b];
    const int64_t pos = position_ids[b] + 1;
    position_ids[b] = pos;
    if (all_ids && pos >= 0 && pos < ld_all) all_ids[b * ld_all + pos] = id;
    if (ids_copy) ids_copy[b] = id;
    if (stage_ids) stage_ids[b] = id;
    if (stage_pos) stage_pos[b] = (int32_t)pos;
}

// argmax (lowest index on ties) + logsumexp per row; one 1024-thread block per row.
template <typename T>
__global__ __launch_bounds__(1024) void argmax_logprob_kernel(const T* __restrict__ logits, int64_t ld,
                                                               int64_t V, int64_t* __restrict__ ids,
                                                               float* __restrict__ logprob) {
    __shared__ float sm[16];
    __shared__ int si[16];
    __shared__ float ss[16];
    const int64_t row = blockIdx.x;
    const T* p = logits + row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int64_t i = threadIdx.x; i < V; i += blockDim.x) {
        float v = (float)p[i];
        if (v > best || (v == best && (int)i < bi)) {
            best = v;
            bi = (int)i;
        }
    }
    // wave reduce (max value, then min index)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        sm[w] = best;
        si[w] = bi;
    }
    __syncthreads();
    best = sm[0];
    bi = si[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
        if (sm[k] > best || (sm[k] == best && si[k] < bi)) {
            best = sm[k];
            bi = si[k];
        }
    }
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < V; i += blockDim.x) s += __expf((float)p[i] - best);
    s = wave_sum(s);
    if (lane == 0) ss[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tot += ss[k];
        ids[row] = bi;
        logprob[row] = -__logf(tot);  // logit[max] - (max + log(sum exp(l - max)))
    }
}

// The same result from SEG workgroups per row (one workgroup streams a row at what one CU can take in, ~5 us for
// 32000 fp32 logits; decode batches have far fewer rows than the chip has CUs): part 1 leaves {max, its lowest index,
// sum exp(l - max)} of each segment, part 2 merges the SEG records of a row.
struct ArgmaxPart {
    float m, s;
    int idx, pad;
};

template <typename T>
__global__ __launch_bounds__(256) void argmax_part_kernel(const T* __restrict__ logits, int64_t ld, int64_t V, int seg_len,
                                                          ArgmaxPart* __restrict__ parts) {
    __shared__ float sm[4];
    __shared__ int si[4];
    __shared__ float ss[4];
    const int64_t row = blockIdx.y;
    const int seg = blockIdx.x;
    const int64_t lo = (int64_t)seg * seg_len, hi = min(V, lo + seg_len);
    const T* p = logits + row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        float v = (float)p[i];
        if (v > best || (v == best && (int)i < bi)) {
            best = v;
            bi = (int)i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        sm[w] = best;
        si[w] = bi;
    }
    __syncthreads();
    best = sm[0];
    bi = si[0];
    for (int k = 1; k < 4; ++k) {
        if (sm[k] > best || (sm[k] == best && si[k] < bi)) {
            best = sm[k];
            bi = si[k];
        }
    }
    float s = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += __expf((float)p[i] - best);  // second pass: cache-resident
    s = wave_sum(s);
    if (lane == 0) ss[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        ArgmaxPart r;
        r.m = best;  // -inf and idx 0x7fffffff for an empty segment: loses every comparison, adds exp(-inf) = 0
        r.s = lo < hi ? ss[0] + ss[1] + ss[2] + ss[3] : 0.f;
        r.idx = bi;
        r.pad = 0;
        parts[row * gridDim.x + seg] = r;
    }
}

__global__ __launch_bounds__(64) void argmax_merge_kernel(const ArgmaxPart* __restrict__ parts, int nseg, int64_t B,
                                                          int64_t* __restrict__ ids, float* __restrict__ logprob) {
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x;
    float m = -INFINITY, s = 0.f;
    int idx = 0x7fffffff;
    if (lane < nseg) {
        const ArgmaxPart r = parts[row * nseg + lane];
        m = r.m, s = r.s, idx = r.idx;
    }
    float best = m;
    int bi = idx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    float t = lane < nseg && s > 0.f ? s * __expf(m - best) : 0.f;
    t = wave_sum(t);
    if (lane == 0) {
        ids[row] = bi;
        logprob[row] = -__logf(t);
    }
}

}  // namespace

extern "C" int tgis_act_mul(const void* gate_up, void* out, int64_t T, int64_t I, int act, int dtype,
                            void* stream) {
    TGIS_CHECK_ARG(gate_up && out && I > 0 && I % 8 == 0, "tgis_act_mul: bad arguments");
    TGIS_CHECK_ARG(act == 1, "tgis_act_mul: only SiLU (act=1) is implemented");
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_act_mul: bad dtype");
    if (T == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    TgisTimedScope timed(TGIS_OP_ACT, st);
    int64_t n = T * (I >> 3);
    dim3 grid((unsigned)cdiv64(n, 256));
    if (dtype == TGIS_F16)
        hipLaunchKernelGGL(act_mul_kernel<f16>, grid, dim3(256), 0, st, (const f16*)gate_up, (f16*)out, T, I);
    else
        hipLaunchKernelGGL(act_mul_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)gate_up, (bf16*)out, T, I);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int tgis_gelu(const void* x, void* out, int64_t n, int tanh_approx, int dtype, void* stream) {
    TGIS_CHECK_ARG(x && out && n >= 0 && n % 8 == 0, "tgis_gelu: n must be a multiple of 8");
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_gelu: bad dtype");
    if (n == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    TgisTimedScope timed(TGIS_OP_ACT, st);
    dim3 grid((unsigned)cdiv64(n / 8, 256));
    if (dtype == TGIS_F16)
        hipLaunchKernelGGL(gelu_kernel<f16>, grid, dim3(256), 0, st, (const f16*)x, (f16*)out, n, tanh_approx);
    else
        hipLaunchKernelGGL(gelu_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, (bf16*)out, n, tanh_approx);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int tgis_embedding(const int64_t* ids, const void* table, const int32_t* positions,
                              const void* pos_table, void* out, int64_t T, int64_t E, int64_t vocab_rows,
                              int64_t id_offset, int dtype, void* stream) {
    TGIS_CHECK_ARG(ids && table && out && E > 0 && E % 8 == 0, "tgis_embedding: bad arguments");
    TGIS_CHECK_ARG(!pos_table || positions, "tgis_embedding: pos_table needs positions");
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_embedding: bad dtype");
    if (T == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TGIS_F16)
        hipLaunchKernelGGL(embedding_kernel<f16>, dim3((unsigned)T), dim3(256), 0, st, ids, (const f16*)table,
                           positions, (const f16*)pos_table, (f16*)out, E, vocab_rows, id_offset);
    else
        hipLaunchKernelGGL(embedding_kernel<bf16>, dim3((unsigned)T), dim3(256), 0, st, ids, (const bf16*)table,
                           positions, (const bf16*)pos_table, (bf16*)out, E, vocab_rows, id_offset);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int tgis_decode_slots(const int32_t* positions, const int32_t* block_tables, int64_t max_pages,
                                 int32_t* slots, int32_t* ctx_lens, int64_t B, void* stream) {
    TGIS_CHECK_ARG(positions && block_tables && slots && ctx_lens && max_pages > 0,
                   "tgis_decode_slots: bad arguments");
    if (B == 0) return TGIS_OK;
    hipLaunchKernelGGL(decode_slots_kernel, dim3((unsigned)cdiv64(B, 64)), dim3(64), 0, (hipStream_t)stream,
                       positions, block_tables, max_pages, slots, ctx_lens, B);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int tgis_decode_advance(const int64_t* ids, int64_t* ids_copy, int64_t* position_ids, int64_t* all_input_ids,
                                   int64_t ld_all, int32_t* cu_seqlens, const int32_t* cu_seqlens_q, int64_t* stage_ids,
                                   int32_t* stage_positions, int64_t B, void* stream) {
    TGIS_CHECK_ARG(ids && position_ids && B >= 0 && (!all_input_ids || ld_all > 0) && (!cu_seqlens || cu_seqlens_q),
                   "tgis_decode_advance: bad arguments");
    if (B == 0) return TGIS_OK;
    hipLaunchKernelGGL(decode_advance_kernel, dim3((unsigned)cdiv64(B + 1, 64)), dim3(64), 0, (hipStream_t)stream, ids,
                       ids_copy, position_ids, all_input_ids, ld_all, cu_seqlens, cu_seqlens_q, stage_ids, stage_positions, B);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int64_t tgis_argmax_scratch_bytes(int64_t B) { return B > 0 ? (int64_t)sizeof(ArgmaxPart) * B * 16 : 0; }

extern "C" int tgis_argmax_logprob(const void* logits, int64_t ld, int64_t B, int64_t V, int logits_f32,
                                   int dtype, int64_t* ids_out, float* logprob_out, void* scratch, int64_t scratch_bytes,
                                   void* stream) {
    TGIS_CHECK_ARG(logits && ids_out && logprob_out && V > 0, "tgis_argmax_logprob: bad arguments");
    if (B == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    TgisTimedScope timed(TGIS_OP_SAMPLE, st);
    TGIS_CHECK_ARG(logits_f32 || dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_argmax_logprob: bad dtype");
    // rows are split over workgroups while the batch leaves most of the chip idle and the caller lent the scratch
    int nseg = (int)std::min<int64_t>(16, 256 / std::max<int64_t>(B, 1));
    while (nseg > 1 && cdiv64(V, nseg) < 1024) --nseg;
    if (scratch && nseg > 1 && scratch_bytes >= (int64_t)sizeof(ArgmaxPart) * B * nseg) {
        const int seg_len = (int)cdiv64(V, nseg);
        ArgmaxPart* parts = (ArgmaxPart*)scratch;
        dim3 grid((unsigned)nseg, (unsigned)B);
        if (logits_f32)
            hipLaunchKernelGGL(argmax_part_kernel<float>, grid, dim3(256), 0, st, (const float*)logits, ld, V, seg_len, parts);
        else if (dtype == TGIS_F16)
            hipLaunchKernelGGL(argmax_part_kernel<f16>, grid, dim3(256), 0, st, (const f16*)logits, ld, V, seg_len, parts);
        else
            hipLaunchKernelGGL(argmax_part_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)logits, ld, V, seg_len, parts);
        hipLaunchKernelGGL(argmax_merge_kernel, dim3((unsigned)B), dim3(64), 0, st, parts, nseg, B, ids_out, logprob_out);
        TGIS_CHECK_LAUNCH();
        return TGIS_OK;
    }
    dim3 grid((unsigned)B), block(1024);
    if (logits_f32)
        hipLaunchKernelGGL(argmax_logprob_kernel<float>, grid, block, 0, st, (const float*)logits, ld, V, ids_out,
                           logprob_out);
    else if (dtype == TGIS_F16)
        hipLaunchKernelGGL(argmax_logprob_kernel<f16>, grid, block, 0, st, (const f16*)logits, ld, V, ids_out,
                           logprob_out);
    else if (dtype == TGIS_BF16)
        hipLaunchKernelGGL(argmax_logprob_kernel<bf16>, grid, block, 0, st, (const bf16*)logits, ld, V, ids_out,
                           logprob_out);
    else {
        tgis_set_error("tgis_argmax_logprob: bad dtype");
        return TGIS_EINVAL;
    }
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}
