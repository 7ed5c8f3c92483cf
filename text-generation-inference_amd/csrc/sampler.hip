// Next-token chooser for a heterogeneous batch in ONE launch: one 1024-thread workgroup per request row.
//
// Replaces, for logits on the GPU, the chain the reference runs as ~10 torch ops per warper plus ~6 per sampled
// request (utils/tokens.py:242-270 `HeterogeneousNextTokenChooser.__call__`, :32-41 `Sampling`, :336-385
// `HeterogeneousSampling`; utils/logits_process.py:93-402 the Heterogeneous* processors), in the same order:
//   EOS mask (min_new_tokens) / length penalty -> repetition penalty -> temperature -> top-k -> top-p -> typical-p
//   -> argmax (greedy rows) or a categorical draw (sampled rows) -> log-softmax of the warped scores at the chosen id.
//
// The row (V fp32 scores, 128 KB at V=32000) is rewritten in place in `scores` and re-read from L2 by every pass; no
// sort is needed anywhere:
//   top-k      radix select of the k-th largest key (4 integer-histogram passes, 8 bits each); ties with the k-th
//              value are kept, as `scores < kth` keeps them in the reference;
//   top-p      the reference removes the ascending-sorted prefix whose cumulative probability is <= 1 - top_p; that
//              prefix is {key <= t*} for the largest t* whose mass A(t*) <= 1 - top_p, found bit by bit (32 masked
//              block sums);
//   typical-p  the reference keeps, in ascending |(-log p) - H| order, everything up to the first element whose
//              prefix mass reaches `mass`; that set is {dist key <= u*} for the largest u* with (mass of keys < u*)
//              < mass — the same search on another key.
// Every block sum has a fixed association order (per-thread strided, wave butterfly, 16 partials in order), so a row's
// result depends only on (its scores, its parameters, its RNG state): not on the batch around it, not on timing.
// Elements exactly tied at a top-p / typical-p boundary are kept or removed together (the reference's sort splits
// such ties arbitrarily).
//
// Sampling is the exponential race the host path used (argmax p_i / E_i, E_i ~ Exp(1)) with a counter-based
// generator: E_i = -log(u_i), u_i from Philox4x32-10 keyed by the request's seed with counter (i, draw offset).
// A request's stream is (seed, offset): it survives concatenate / prune by copying two integers.
#include "common.h"

namespace {

constexpr int NT = 1024;
constexpr int NW = NT / 64;

struct SampleArgs {
    const float* logits;
    int64_t ld_logits;
    float* scores;
    int64_t ld_scores;
    int V;
    const float* temperature;   // [B] or null
    const int* top_k;           // [B] or null; 0 = off
    const float* top_p_cut;     // [B] or null: 1 - top_p as the host rounded it; <= 0 = off
    const float* typical_p;     // [B] or null; >= 1 = off
    const float* rep_penalty;   // [B] or null; 1 = off
    const int64_t* input_ids;   // [B, L] ids seen so far (padding included, as in the reference)
    int64_t ld_ids;
    int L;
    int exclude_id;             // id whose score the repetition penalty leaves alone, or -1
    const float* eos_adjust;    // [B, 2] (mode, factor) or null; mode 1: -inf, mode 2: s + |s| * factor
    int eos_id;
    const int* do_sample;       // [B] or null (all greedy)
    uint64_t* rng;              // [B, 2] (seed, offset); offset is advanced for sampled rows
    int64_t* next_ids;
    float* next_logprob;
    float* lse;
};

__device__ __forceinline__ uint32_t order_key(float x) {  // unsigned order == float order
    uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Block {
    float* fbuf;   // [NW]
    int* ibuf;     // [NW]
    int lane, wave;

    __device__ float sum(float v) {
        v = wave_sum(v);
        __syncthreads();  // the previous reduction's readers are done
        if (lane == 0) fbuf[wave] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) t += fbuf[k];
        return t;
    }
    __device__ float max(float v) {
        v = wave_max(v);
        __syncthreads();
        if (lane == 0) fbuf[wave] = v;
        __syncthreads();
        float t = fbuf[0];
#pragma unroll
        for (int k = 1; k < NW; ++k) t = fmaxf(t, fbuf[k]);
        return t;
    }
    // (value, index) of the maximum, lowest index on ties
    __device__ void argmax(float& v, int& i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(v, o, 64);
            int oi = __shfl_xor(i, o, 64);
            if (ov > v || (ov == v && oi < i)) {
                v = ov;
                i = oi;
            }
        }
        __syncthreads();
        if (lane == 0) {
            fbuf[wave] = v;
            ibuf[wave] = i;
        }
        __syncthreads();
        v = fbuf[0];
        i = ibuf[0];
#pragma unroll
        for (int k = 1; k < NW; ++k)
            if (fbuf[k] > v || (fbuf[k] == v && ibuf[k] < i)) {
                v = fbuf[k];
                i = ibuf[k];
            }
    }
};

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0;
    c[1] = lo1;
    c[2] = n2;
    c[3] = lo0;
}

// first word of Philox4x32-10(counter = (i, 0, offset_lo, offset_hi), key = seed)
__device__ __forceinline__ uint32_t philox_u32(uint64_t seed, uint64_t offset, uint32_t i) {
    uint32_t c[4] = {i, 0u, (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c[0];
}

__global__ __launch_bounds__(NT) void warp_sample_kernel(SampleArgs a) {
    __shared__ float fbuf[NW];
    __shared__ int ibuf[NW];
    __shared__ int hist[256];
    __shared__ uint32_t bc[2];
    const int b = blockIdx.x, tid = threadIdx.x, V = a.V;
    Block blk{fbuf, ibuf, tid & 63, tid >> 6};
    const float* in = a.logits + (int64_t)b * a.ld_logits;
    float* s = a.scores + (int64_t)b * a.ld_scores;
    const float NEG_INF = -INFINITY;

    // ---- EOS adjustment, repetition penalty, temperature ---------------------------------------------------------
    const float T = a.temperature ? a.temperature[b] : 1.0f;
    int eos_mode = 0;
    float eos_factor = 0.f;
    if (a.eos_adjust) {
        eos_mode = (int)a.eos_adjust[2 * b];
        eos_factor = a.eos_adjust[2 * b + 1];
    }
    auto adjusted = [&](int i) {
        float x = in[i];
        if (eos_mode && i == a.eos_id) x = eos_mode == 1 ? NEG_INF : x + fabsf(x) * eos_factor;
        return x;
    };
    for (int i = tid; i < V; i += NT) s[i] = adjusted(i) / T;
    const float pen = a.rep_penalty ? a.rep_penalty[b] : 1.0f;
    if (pen != 1.0f && a.input_ids) {
        __syncthreads();
        const int64_t* ids = a.input_ids + (int64_t)b * a.ld_ids;
        for (int j = tid; j < a.L; j += NT) {  // duplicates write the same value: each id is penalised once
            int64_t id = ids[j];
            if (id < 0 || id >= V || id == a.exclude_id) continue;
            float x = adjusted((int)id);
            x = x < 0.f ? x * pen : x / pen;
            s[id] = x / T;
        }
    }
    __syncthreads();

    // ---- top-k: radix select of the k-th largest key ---------------------------------------------------------------
    int k = a.top_k ? a.top_k[b] : 0;
    if (k > 0 && k < V) {
        uint32_t prefix = 0, known = 0;
        int want = k;  // rank (from the top) still to resolve inside the current prefix
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += NT) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < V; i += NT) {
                uint32_t key = order_key(s[i]);
                if ((key & known) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int bin = 255, left = want;
                for (; bin > 0; --bin) {
                    if (hist[bin] >= left) break;
                    left -= hist[bin];
                }
                bc[0] = (uint32_t)bin;
                bc[1] = (uint32_t)left;
            }
            __syncthreads();
            prefix |= bc[0] << shift;
            known |= 255u << shift;
            want = (int)bc[1];
            __syncthreads();
        }
        for (int i = tid; i < V; i += NT)
            if (order_key(s[i]) < prefix) s[i] = NEG_INF;
        __syncthreads();
    }

    // ---- top-p -------------------------------------------------------------------------------------------------
    const float cut = a.top_p_cut ? a.top_p_cut[b] : 0.0f;
    if (cut > 0.0f) {
        float m = NEG_INF;
        for (int i = tid; i < V; i += NT) m = fmaxf(m, s[i]);
        m = blk.max(m);
        float z = 0.f;
        for (int i = tid; i < V; i += NT) z += expf(s[i] - m);
        z = blk.sum(z);
        const float inv_z = 1.0f / z;
        uint32_t t = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = t | (1u << bit);
            float acc = 0.f;
            for (int i = tid; i < V; i += NT) {
                float x = s[i];
                if (order_key(x) <= cand) acc += expf(x - m) * inv_z;
            }
            if (blk.sum(acc) <= cut) t = cand;
        }
        const uint32_t top = order_key(m);  // min_tokens_to_keep = 1
        if (t >= top) t = top - 1;
        __syncthreads();
        for (int i = tid; i < V; i += NT)
            if (order_key(s[i]) <= t) s[i] = NEG_INF;
        __syncthreads();
    }

    // ---- typical-p ---------------------------------------------------------------------------------------------
    const float mass = a.typical_p ? a.typical_p[b] : 1.0f;
    if (mass < 1.0f) {
        float m = NEG_INF;
        for (int i = tid; i < V; i += NT) m = fmaxf(m, s[i]);
        m = blk.max(m);
        float z = 0.f;
        for (int i = tid; i < V; i += NT) z += expf(s[i] - m);
        z = blk.sum(z);
        const float log_z = m + logf(z), inv_z = 1.0f / z;
        float h = 0.f;
        for (int i = tid; i < V; i += NT) {
            float x = s[i];
            if (x != NEG_INF) {  // nansum: 0 * -inf terms are skipped
                float lp = x - log_z;
                h -= expf(lp) * lp;
            }
        }
        const float ent = blk.sum(h);
        auto dist_key = [&](float x) { return __float_as_uint(fabsf((log_z - x) - ent)); };  // >= 0: bits are ordered
        uint32_t u = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = u | (1u << bit);
            float acc = 0.f;
            for (int i = tid; i < V; i += NT) {
                float x = s[i];
                if (dist_key(x) < cand) acc += expf(x - m) * inv_z;
            }
            if (blk.sum(acc) < mass) u = cand;
        }
        __syncthreads();
        for (int i = tid; i < V; i += NT)
            if (dist_key(s[i]) > u) s[i] = NEG_INF;
        __syncthreads();
    }

    // ---- choice + log-softmax of the warped scores at the chosen id ----------------------------------------------
    float m = NEG_INF;
    int mi = 0x7fffffff;
    for (int i = tid; i < V; i += NT) {
        float x = s[i];
        if (x > m || (x == m && i < mi)) {
            m = x;
            mi = i;
        }
    }
    blk.argmax(m, mi);
    if (mi >= V) mi = 0;  // a row of NaNs: no comparison ever succeeded
    float z = 0.f;
    for (int i = tid; i < V; i += NT) z += expf(s[i] - m);
    z = blk.sum(z);
    const float log_z = m + logf(z);
    int chosen = mi;
    float chosen_score = m;
    if (a.do_sample && a.do_sample[b]) {
        const uint64_t seed = a.rng[2 * b], offset = a.rng[2 * b + 1];
        float best = NEG_INF;
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += NT) {
            float x = s[i];
            if (x == NEG_INF) continue;
            float uu = ((float)philox_u32(seed, offset, (uint32_t)i) + 0.5f) * 2.3283064365386963e-10f;  // (0, 1]
            uu = fminf(uu, 0.99999994f);
            float g = (x - m) - logf(-logf(uu));  // log(p_i / E_i) up to a constant
            if (g > best || (g == best && i < bi)) {
                best = g;
                bi = i;
            }
        }
        blk.argmax(best, bi);
        if (bi >= V) bi = mi;
        chosen = bi;
        chosen_score = s[bi];
        if (tid == 0) a.rng[2 * b + 1] = offset + 1;
    }
    if (tid == 0) {
        a.next_ids[b] = chosen;
        a.next_logprob[b] = chosen_score - log_z;
        a.lse[b] = log_z;
    }
}


// The same chooser with the row in registers (V <= NPT * 1024): after the EOS / repetition-penalty / temperature pass has
// written the row once (the penalty is a scatter), thread t loads elements t, t + 1024, ... — the very elements the loops
// above hand it — and every later pass (4 histogram passes, 2 x 32 masked sums, maxima, sums, the draw) runs on those
// registers; the probabilities of the top-p / typical-p searches are computed once per search instead of once per bit.
// Same per-thread element order, same expressions, same block reductions: bit-identical to warp_sample_kernel, which stays
// for larger vocabularies.  (The global form re-read the 128 KB row from L2 ~80 times: 0.4 ms per step at V = 32000.)
template <int NPT>
__global__ __launch_bounds__(NT) void warp_sample_reg_kernel(SampleArgs a) {
    __shared__ float fbuf[NW];
    __shared__ int ibuf[NW];
    __shared__ int hist[256];
    __shared__ uint32_t bc[2];
    const int b = blockIdx.x, tid = threadIdx.x, V = a.V;
    Block blk{fbuf, ibuf, tid & 63, tid >> 6};
    const float* in = a.logits + (int64_t)b * a.ld_logits;
    float* s = a.scores + (int64_t)b * a.ld_scores;
    const float NEG_INF = -INFINITY;

    const float T = a.temperature ? a.temperature[b] : 1.0f;
    int eos_mode = 0;
    float eos_factor = 0.f;
    if (a.eos_adjust) {
        eos_mode = (int)a.eos_adjust[2 * b];
        eos_factor = a.eos_adjust[2 * b + 1];
    }
    auto adjusted = [&](int i) {
        float x = in[i];
        if (eos_mode && i == a.eos_id) x = eos_mode == 1 ? NEG_INF : x + fabsf(x) * eos_factor;
        return x;
    };
    float x[NPT];
    const float pen = a.rep_penalty ? a.rep_penalty[b] : 1.0f;
    const bool scatter = pen != 1.0f && a.input_ids;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int i = tid + j * NT;
        x[j] = i < V ? adjusted(i) / T : NEG_INF;
    }
    if (scatter) {
        // the penalised entries go through the row in memory (any thread may hit any id), then come back into registers
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int i = tid + j * NT;
            if (i < V) s[i] = x[j];
        }
        __syncthreads();
        const int64_t* ids = a.input_ids + (int64_t)b * a.ld_ids;
        for (int j = tid; j < a.L; j += NT) {  // duplicates write the same value: each id is penalised once
            int64_t id = ids[j];
            if (id < 0 || id >= V || id == a.exclude_id) continue;
            float v = adjusted((int)id);
            v = v < 0.f ? v * pen : v / pen;
            s[id] = v / T;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int i = tid + j * NT;
            if (i < V) x[j] = s[i];
        }
    }

    // ---- top-k ----------------------------------------------------------------------------------------------------
    int k = a.top_k ? a.top_k[b] : 0;
    if (k > 0 && k < V) {
        uint32_t prefix = 0, known = 0;
        int want = k;
#pragma unroll 1
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += NT) hist[i] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                if (tid + j * NT < V) {
                    uint32_t key = order_key(x[j]);
                    if ((key & known) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
                }
            }
            __syncthreads();
            if (tid == 0) {
                int bin = 255, left = want;
                for (; bin > 0; --bin) {
                    if (hist[bin] >= left) break;
                    left -= hist[bin];
                }
                bc[0] = (uint32_t)bin;
                bc[1] = (uint32_t)left;
            }
            __syncthreads();
            prefix |= bc[0] << shift;
            known |= 255u << shift;
            want = (int)bc[1];
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            if (order_key(x[j]) < prefix) x[j] = NEG_INF;
    }

    // ---- top-p ----------------------------------------------------------------------------------------------------
    const float cut = a.top_p_cut ? a.top_p_cut[b] : 0.0f;
    if (cut > 0.0f) {
        float m = NEG_INF;
#pragma unroll
        for (int j = 0; j < NPT; ++j) m = fmaxf(m, x[j]);
        m = blk.max(m);
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            if (tid + j * NT < V) z += expf(x[j] - m);
        z = blk.sum(z);
        const float inv_z = 1.0f / z;
        float p[NPT];  // (the keys are three operations away from x: recomputed per bit, the registers go to p)
#pragma unroll
        for (int j = 0; j < NPT; ++j) p[j] = tid + j * NT < V ? expf(x[j] - m) * inv_z : 0.f;
        uint32_t t = 0;
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = t | (1u << bit);
            float acc = 0.f;
            uint32_t zb = 0;
            asm volatile("" : "+v"(zb));  // an opaque zero: keeps the 32 keys from being hoisted out of the bit loop (registers)
#pragma unroll
            for (int j = 0; j < NPT; ++j)
                if (order_key(__uint_as_float(__float_as_uint(x[j]) | zb)) <= cand) acc += p[j];  // p is 0 beyond the row
            if (blk.sum(acc) <= cut) t = cand;
        }
        const uint32_t top = order_key(m);  // min_tokens_to_keep = 1
        if (t >= top) t = top - 1;
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            if (order_key(x[j]) <= t) x[j] = NEG_INF;
    }

    // ---- typical-p ------------------------------------------------------------------------------------------------
    const float mass = a.typical_p ? a.typical_p[b] : 1.0f;
    if (mass < 1.0f) {
        float m = NEG_INF;
#pragma unroll
        for (int j = 0; j < NPT; ++j) m = fmaxf(m, x[j]);
        m = blk.max(m);
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            if (tid + j * NT < V) z += expf(x[j] - m);
        z = blk.sum(z);
        const float log_z = m + logf(z), inv_z = 1.0f / z;
        float h = 0.f;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            if (tid + j * NT < V && x[j] != NEG_INF) {  // nansum: 0 * -inf terms are skipped
                float lp = x[j] - log_z;
                h -= expf(lp) * lp;
            }
        }
        const float ent = blk.sum(h);
        auto dist_key = [&](float v) { return __float_as_uint(fabsf((log_z - v) - ent)); };  // >= 0: bits are ordered
        float p[NPT];
#pragma unroll
        for (int j = 0; j < NPT; ++j) p[j] = tid + j * NT < V ? expf(x[j] - m) * inv_z : 0.f;
        uint32_t u = 0;
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = u | (1u << bit);
            float acc = 0.f;
            uint32_t zb = 0;
            asm volatile("" : "+v"(zb));
#pragma unroll
            for (int j = 0; j < NPT; ++j)
                if (dist_key(__uint_as_float(__float_as_uint(x[j]) | zb)) < cand) acc += p[j];
            if (blk.sum(acc) < mass) u = cand;
        }
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            if (dist_key(x[j]) > u) x[j] = NEG_INF;
    }

    // the warped scores are an output (top-n tokens and ranks read them); the chosen score is read back from them
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int i = tid + j * NT;
        if (i < V) s[i] = x[j];
    }

    // ---- choice + log-softmax of the warped scores at the chosen id ----------------------------------------------
    float m = NEG_INF;
    int mi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int i = tid + j * NT;
        if (i < V && (x[j] > m || (x[j] == m && i < mi))) {
            m = x[j];
            mi = i;
        }
    }
    blk.argmax(m, mi);
    if (mi >= V) mi = 0;  // a row of NaNs: no comparison ever succeeded
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < NPT; ++j)
        if (tid + j * NT < V) z += expf(x[j] - m);
    z = blk.sum(z);
    const float log_z = m + logf(z);
    int chosen = mi;
    float chosen_score = m;
    if (a.do_sample && a.do_sample[b]) {
        const uint64_t seed = a.rng[2 * b], offset = a.rng[2 * b + 1];
        float best = NEG_INF;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int i = tid + j * NT;
            if (i >= V || x[j] == NEG_INF) continue;
            float uu = ((float)philox_u32(seed, offset, (uint32_t)i) + 0.5f) * 2.3283064365386963e-10f;  // (0, 1]
            uu = fminf(uu, 0.99999994f);
            float g = (x[j] - m) - logf(-logf(uu));  // log(p_i / E_i) up to a constant
            if (g > best || (g == best && i < bi)) {
                best = g;
                bi = i;
            }
        }
        blk.argmax(best, bi);  // (its barriers also order the row's stores above before the read below)
        if (bi >= V) bi = mi;
        chosen = bi;
        chosen_score = s[bi];
        if (tid == 0) a.rng[2 * b + 1] = offset + 1;
    }
    if (tid == 0) {
        a.next_ids[b] = chosen;
        a.next_logprob[b] = chosen_score - log_z;
        a.lse[b] = log_z;
    }
}

}  // namespace

extern "C" int tgis_warp_sample(const float* logits, int64_t ld_logits, float* scores, int64_t ld_scores, int64_t B,
                                int64_t V, const float* temperature, const int* top_k, const float* top_p_cut,
                                const float* typical_p, const float* rep_penalty, const int64_t* input_ids,
                                int64_t ld_ids, int64_t L, int64_t exclude_id, const float* eos_adjust, int64_t eos_id,
                                const int* do_sample, uint64_t* rng, int64_t* next_ids, float* next_logprob, float* lse,
                                void* stream) {
    TGIS_CHECK_ARG(logits && scores && next_ids && next_logprob && lse, "tgis_warp_sample: null buffer");
    TGIS_CHECK_ARG(V > 0 && V < (1ll << 31) && ld_logits >= V && ld_scores >= V, "tgis_warp_sample: bad row geometry");
    TGIS_CHECK_ARG(!do_sample || rng, "tgis_warp_sample: sampled rows need their (seed, offset) states");
    TGIS_CHECK_ARG(!rep_penalty || (input_ids && L >= 0 && L < (1ll << 31)),
                   "tgis_warp_sample: a repetition penalty needs the ids seen so far");
    TGIS_CHECK_ARG(!eos_adjust || (eos_id >= 0 && eos_id < V), "tgis_warp_sample: EOS adjustment without a valid EOS id");
    if (B == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    TgisTimedScope timed(TGIS_OP_SAMPLE, st);
    SampleArgs a{logits, ld_logits, scores, ld_scores, (int)V, temperature, top_k, top_p_cut, typical_p, rep_penalty,
                 input_ids, ld_ids, (int)L, (int)exclude_id, eos_adjust, (int)eos_id, do_sample, rng, next_ids,
                 next_logprob, lse};
    static const bool no_reg = getenv("TGIS_SAMPLER_GLOBAL_ROWS") != nullptr;  // A / B and test hook
    if (V <= 32 * NT && !no_reg)
        hipLaunchKernelGGL(warp_sample_reg_kernel<32>, dim3((unsigned)B), dim3(NT), 0, st, a);
    else
        hipLaunchKernelGGL(warp_sample_kernel, dim3((unsigned)B), dim3(NT), 0, st, a);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}
