// Persistent "decode tail" of a Llama layer for gfx950: everything between two attention launches in ONE launch.
//
//   o_proj GEMM -> | -> add + RMSNorm -> | -> gate_up GEMM (SiLU*up epilogue) -> | -> down GEMM -> | -> add + RMSNorm
//   -> | -> qkv GEMM of the NEXT layer -> | -> RoPE + KV-cache write of the next layer            ( | = grid barrier)
//
// It replaces seven dependent launches of the round-1 decode step (flash_llama_modeling.py:285-297,368-385,332-335 of
// the reference: o_proj, post_attention_layernorm, gate_up/down, the next layer's input_layernorm, query_key_value,
// rotary + cache write) with the same arithmetic — the phases call the very device code of gptq.hip, norm.hip and
// rope_kv.hip semantics, so the results are bit-identical to the multi-launch path.
//
// Why it pays on MI355X (tools/floor/xcdbar.hip, profiles/r02_barrier.md): a grid barrier over 256 resident
// workgroups costs 1.8 us when it carries no fences — two-level arrival (8 group counters -> one top counter -> 8
// generation words polled relaxed) with the inter-workgroup data travelling as sc1 write-through stores (drained with
// vmcnt(0) before arriving) and sc1 loads — against 3-5 us of launch, dispatch ramp and drain per kernel boundary.
//
// One workgroup of 12 waves per CU, all resident (the launcher checks the occupancy); a phase's units are dealt to the
// workgroups round-robin; waves a unit does not use idle at the next workgroup barrier.  Every spin is bounded: a
// barrier that cannot complete (a workgroup not resident) sets an error word instead of hanging the device.
#include <utility>
#include <vector>
#include "common.h"
#include "../include/tgis_experiments.h"
#include "dense_gemm_body.h"
#include "gptq_gemm_body.h"

namespace {

using dense::DenseArgs;
using gptq::GemmArgs;
using gptq::GemmPlan;

constexpr unsigned SPIN_LIMIT = 1u << 22;

#define RLX_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RLX_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RLX_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// Monotonic across launches (a launch reads the generation it starts from), one 128-byte line per hot word.
struct GridBar {
    unsigned top, pad0[31];
    unsigned grp[8][32];
    unsigned gen[8][32];
    unsigned err, pad1[31];
};

struct NormPhase {
    const float* slabs;  // [S][32][slab_ld] split-K partial sums of the GEMM before
    int S;
    int64_t slab_ld;
    const void* xbias;     // bias of that GEMM or nullptr          (T = the model dtype, f16 or bf16)
    const void* residual;  // T [rows, hidden] or nullptr
    const void* weight;
    void* y;
    void* res_out;
    int rows, hidden;
    float eps;
};

struct RopePhase {
    const float* slabs;
    int S;
    int64_t slab_ld;
    const void* bias;
    void* qkv;  // [T, ld] out: rotated q, k and v
    int64_t ld;
    const void* cosb;
    const void* sinb;
    const int32_t* positions;
    const int32_t* slots;
    void* kpool;
    void* vpool;
    int T, H, Hkv, D, rot;
};

template <class GArgs>
struct TailArgsT {
    GArgs g[4];          // o_proj, gate_up, down, qkv of the next layer
    int tw[4];           // TN * 10 + WK of each plan
    int gx[4], gy[4];    // column blocks, k splits
    NormPhase n[2];
    RopePhase r;
    int phases;          // 5: stop after the second norm (last layer), 7: all
    GridBar* bar;
    long long* trace;    // debug: [workgroups][16] s_memrealtime stamps (100 MHz) at the phase edges, or nullptr
};
using TailArgs = TailArgsT<GemmArgs>;
using DenseTailArgs = TailArgsT<DenseArgs>;

// ---- grid barrier ------------------------------------------------------------------------------------------------
struct BarCtx {
    unsigned epoch;   // generation of the barrier this workgroup arrives at next (thread 0)
    unsigned gsize;   // workgroups in this workgroup's group
    unsigned ngroups;
};

__device__ __forceinline__ BarCtx bar_init(GridBar* b) {
    BarCtx c;
    const unsigned g = blockIdx.x & 7u;
    c.ngroups = min(8u, gridDim.x);
    c.gsize = (gridDim.x - g + 7u) / 8u;
    c.epoch = 0;
    if (threadIdx.x == 0) c.epoch = RLX_LOAD(&b->gen[g][0]);  // all workgroups of a launch read the same value
    return c;
}

// `between` runs on every wave after the workgroup has arrived and before it waits: loads issued there (the next
// phase's weights) are in flight while the barrier completes.
template <class F>
__device__ __forceinline__ void grid_sync(GridBar* b, BarCtx& c, F&& between) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have left the CU
    __syncthreads();
    const unsigned e = c.epoch + 1u;
    const unsigned g = blockIdx.x & 7u;
    if (threadIdx.x == 0) {
        c.epoch = e;
        const unsigned old = RLX_ADD(&b->grp[g][0], 1u);
        if (old + 1u == e * c.gsize) {  // last of the group (arithmetic mod 2^32 on both sides)
            const unsigned t = RLX_ADD(&b->top, 1u);
            if (t + 1u == e * c.ngroups) {
                for (unsigned gg = 0; gg < c.ngroups; ++gg) RLX_STORE(&b->gen[gg][0], e);
            }
        }
    }
    between();
    if (threadIdx.x == 0) {
        for (unsigned spins = 0; (int)(RLX_LOAD(&b->gen[g][0]) - e) < 0; ++spins) {
            __builtin_amdgcn_s_sleep(1);
            if (spins > SPIN_LIMIT) {
                RLX_STORE(&b->err, 1u);
                break;
            }
        }
    }
    // execution barrier only: __syncthreads() would also wait (vmcnt) for the loads `between` has just issued
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- sc1 (L1-bypassing / write-through) 16-byte accesses ---------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ u32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, int64_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)byte_off, 0, 16);
}
__device__ __forceinline__ void st_sc1(u32x4 v, __amdgpu_buffer_rsrc_t r, int64_t byte_off) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (uint32_t)byte_off, 0, 16);
}

// 8 consecutive columns of sum_s slabs[s][row][col..] (+ bias), rounded to f16 — the arithmetic of common.h's
// sum_slabs8 + norm.hip / rope_kv.hip's rounding (fixed slab order), with all loads of the row issued first.
template <typename T, int SB>
__device__ __forceinline__ typename VecT<T>::x8 sum_slabs_t(__amdgpu_buffer_rsrc_t r, int64_t elem_off, int64_t stride,
                                                            int S, const T* bias, int col) {
    u32x4 l[SB], h[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const int64_t o = (elem_off + (int64_t)min(s, S - 1) * stride) * 4;
        l[s] = ld_sc1(r, o);
        h[s] = ld_sc1(r, o + 16);
    }
    f32x4 lo = __builtin_bit_cast(f32x4, l[0]), hi = __builtin_bit_cast(f32x4, h[0]);
#pragma unroll
    for (int s = 1; s < SB; ++s) {
        if (s < S) {
            lo += __builtin_bit_cast(f32x4, l[s]);
            hi += __builtin_bit_cast(f32x4, h[s]);
        }
    }
    for (int s = SB; s < S; ++s) {
        const int64_t o = (elem_off + (int64_t)s * stride) * 4;
        lo += __builtin_bit_cast(f32x4, ld_sc1(r, o));
        hi += __builtin_bit_cast(f32x4, ld_sc1(r, o + 16));
    }
    typename VecT<T>::x8 a;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float b0 = bias ? to_f32(bias[col + e]) : 0.f, b1 = bias ? to_f32(bias[col + 4 + e]) : 0.f;
        a[e] = from_f32<T>(lo[e] + b0);
        a[e + 4] = from_f32<T>(hi[e] + b1);
    }
    return a;
}
template <typename T>
__device__ __forceinline__ typename VecT<T>::x8 sum_slabs_t(__amdgpu_buffer_rsrc_t r, int64_t elem_off, int64_t stride,
                                                            int S, const T* bias, int col) {
    if (S <= 1) return sum_slabs_t<T, 1>(r, elem_off, stride, S, bias, col);
    if (S <= 2) return sum_slabs_t<T, 2>(r, elem_off, stride, S, bias, col);
    if (S <= 4) return sum_slabs_t<T, 4>(r, elem_off, stride, S, bias, col);
    return sum_slabs_t<T, 8>(r, elem_off, stride, S, bias, col);
}

// ---- phases --------------------------------------------------------------------------------------------------------
constexpr int TAIL_RING = 4;  // measured: an 8-deep ring only lengthens the barriers it is filled in (76 vs 69 us per launch)
constexpr int PF = gptq::UNIT_PREFETCH, RUN = gptq::UNIT_RUN, FULL = gptq::UNIT_FULL;

// The two GEMM families the tail is built for.  THREADS: size of the persistent workgroup (every plan's TN * WK waves
// must fit); ACT1 / ACT2: the unit flavour of gate_up / down (both families: gate_up carries the SiLU*up epilogue, its
// image has interleaved gate / up columns).
struct GptqFamily {
    using T = f16;
    using Args = GemmArgs;
    using Ring = gptq::WeightRing<TAIL_RING>;
    static constexpr int THREADS = 768;  // 12 waves: every plan of the 7B-class shapes (TN x WK <= 12), 168 VGPRs, no spills
    static constexpr int ACT1 = 2, ACT2 = 0;
    template <int TN, int WK, int ACT, int MODE>
    static __device__ __forceinline__ void unit(const Args& a, int ntg, int split, unsigned char* smem, int ub, Ring& ring) {
        gptq::gptq_gemm_unit<TN, WK, ACT, true, false, 1, true, TAIL_RING, MODE>(a, ntg, split, 0, smem, ub, ring);
    }
};
template <typename TT>
struct DenseFamily {
    using T = TT;
    using Args = DenseArgs;
    using Ring = dense::DenseRing<TT>;
    static constexpr int THREADS = 512;  // 8 waves: the 1B-class shapes plan TN x WK <= 8
    static constexpr int ACT1 = 2, ACT2 = 0;
    template <int TN, int WK, int ACT, int MODE>
    static __device__ __forceinline__ void unit(const Args& a, int ntg, int split, unsigned char* smem, int ub, Ring& ring) {
        dense::dense_gemm_unit<TT, TN, WK, ACT, 1, true, MODE>(a, ntg, split, 0, smem, ub, ring);
    }
};

// A GEMM phase with the plan TW = TN * 10 + WK.  MODE UNIT_PREFETCH: fill the weight ring of this workgroup's unit
// (called inside the grid barrier that precedes the phase); UNIT_RUN: run the unit on the pre-filled ring; UNIT_FULL: both.
template <class F, int ACT, int MODE, int TW>
__device__ __forceinline__ void gemm_phase(const typename F::Args& a, const int gx, const int gy, unsigned char* smem,
                                           const int wave, int& ub, typename F::Ring& ring) {
    constexpr int T = TW / 10, W = TW % 10;
    static_assert(T * W * 64 <= F::THREADS && (W == 2 || W == 4) && T >= 2 && T <= 4, "plan does not fit the workgroup");
    const int u = blockIdx.x;  // one unit per workgroup and phase (the launcher checks units <= workgroups)
    if (u >= gx * gy) return;
    const int ntg = u % gx, split = u / gx;
    if (wave < T * W) F::template unit<T, W, ACT, MODE>(a, ntg, split, smem, ub, ring);
    if (MODE != PF) ub += gptq::unit_barriers(W) * T * W;
}

template <int THREADS>
__device__ __forceinline__ float block_sum_all(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < THREADS / 64; ++k) t += sh[k];
    return t;
}

// res = T(sum of slabs (+bias)) + residual; y = res * rsqrt(mean(res^2) + eps) * weight: norm.hip's RMS/PARTIAL
// arithmetic, one row per workgroup, 8 elements per thread (hidden <= THREADS * 8; the launcher checks).
template <typename T>
struct NormPre {                 // what a norm phase can load before the barrier in front of it
    typename VecT<T>::x8 wv, res;
};

// rows <= workgroups (one row per workgroup): the norm weight and the residual row do not depend on the phase before
template <typename T>
__device__ __forceinline__ void norm_prefetch(const NormPhase& p, NormPre<T>& pre) {
    using V8 = typename VecT<T>::x8;
    const int row = blockIdx.x;
    if (row >= p.rows) return;
    __amdgpu_buffer_rsrc_t rr = rsrc_of(p.residual);
    const int c = threadIdx.x;
    if (c < (p.hidden >> 3)) {
        pre.wv = ld16<V8>(reinterpret_cast<const T*>(p.weight) + c * 8);
        if (p.residual) pre.res = __builtin_bit_cast(V8, ld_sc1(rr, ((int64_t)row * p.hidden + c * 8) * 2));
    }
}

template <typename T, int THREADS>
__device__ __forceinline__ void run_norm(const NormPhase& p, unsigned char* smem, const NormPre<T>& pre) {
    using V8 = typename VecT<T>::x8;
    float* sh = reinterpret_cast<float*>(smem);
    const int nchunk = p.hidden >> 3;
    __amdgpu_buffer_rsrc_t rs = rsrc_of(p.slabs), ry = rsrc_of(p.y), ro = rsrc_of(p.res_out);
    for (int row = blockIdx.x; row < p.rows; row += gridDim.x) {
        float v[8];
        float s2 = 0.f;
        const int c = threadIdx.x;
        if (c < nchunk) {
            const V8 a = sum_slabs_t<T>(rs, ((int64_t)(row >> 5) * p.S * 32 + (row & 31)) * p.slab_ld + c * 8,
                                        32 * p.slab_ld, p.S, reinterpret_cast<const T*>(p.xbias), c * 8);
            V8 o;
            if (p.residual) {
                const V8 b = pre.res;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = to_f32(a[e]) + to_f32(b[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = to_f32(a[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = from_f32<T>(v[e]);
                s2 += v[e] * v[e];
            }
            st_sc1(__builtin_bit_cast(u32x4, o), ro, ((int64_t)row * p.hidden + c * 8) * 2);
        }
        const float rstd = rsqrtf(block_sum_all<THREADS>(s2, sh) / p.hidden + p.eps);
        if (c < nchunk) {
            V8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = from_f32<T>((v[e] - 0.f) * rstd * to_f32(pre.wv[e]));
            st_sc1(__builtin_bit_cast(u32x4, o), ry, ((int64_t)row * p.hidden + c * 8) * 2);
        }
    }
}

__device__ __forceinline__ int64_t k_off(int tok, int d, int D) {
    return ((int64_t)(((tok >> 4) * (D >> 3) + (d >> 3)) * 16 + (tok & 15)) << 3) + (d & 7);
}
__device__ __forceinline__ int v_col(int tok) {
    const int i = tok & 15;
    return (i >> 2) * 8 + (tok >> 4) * 4 + (i & 3);
}

// rope_kv.hip's decode form on partial input, items spread over the whole grid.  Its outputs (rotated qkv, KV pages)
// are read by the NEXT launch (attention), so they are plain stores.
// first item of a thread: the cache slot and the rotary factors do not depend on the qkv GEMM
template <typename T>
struct RopePre {
    int slot;
    typename VecT<T>::x8 c, s;
};
template <typename T, int THREADS>
__device__ __forceinline__ void rope_prefetch(const RopePhase& p, RopePre<T>& pre) {
    using V8 = typename VecT<T>::x8;
    const T* cosb = reinterpret_cast<const T*>(p.cosb);
    const T* sinb = reinterpret_cast<const T*>(p.sinb);
    const int c8 = p.D >> 3, rh8 = p.rot >> 4;
    const int per_tok = (p.H + 2 * p.Hkv) * c8;
    const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (idx >= (int64_t)p.T * per_tok) return;
    const int64_t t = idx / per_tok;
    const int it = (int)(idx - t * per_tok);
    const int j = it % c8;
    pre.slot = p.slots[t];
    if (cosb != nullptr && j < rh8) {
        pre.c = ld16<V8>(cosb + (int64_t)p.positions[t] * (p.rot >> 1) + j * 8);
        pre.s = ld16<V8>(sinb + (int64_t)p.positions[t] * (p.rot >> 1) + j * 8);
    }
}

template <typename T, int THREADS>
__device__ __forceinline__ void run_rope(const RopePhase& p, const RopePre<T>& pre) {
    using V8 = typename VecT<T>::x8;
    const T* cosb = reinterpret_cast<const T*>(p.cosb);
    const T* sinb = reinterpret_cast<const T*>(p.sinb);
    const T* bias = reinterpret_cast<const T*>(p.bias);
    T* kpool = reinterpret_cast<T*>(p.kpool);
    T* vpool = reinterpret_cast<T*>(p.vpool);
    const int c8 = p.D >> 3, rh8 = p.rot >> 4;
    const int per_tok = (p.H + 2 * p.Hkv) * c8;
    const int64_t total = (int64_t)p.T * per_tok;
    __amdgpu_buffer_rsrc_t rs = rsrc_of(p.slabs);
    const int64_t first = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    for (int64_t idx = first; idx < total; idx += (int64_t)gridDim.x * THREADS) {
        const int64_t t = idx / per_tok;
        const int it = (int)(idx - t * per_tok);
        const int head = it / c8, j = it - head * c8;
        T* hp = reinterpret_cast<T*>(p.qkv) + t * p.ld + head * p.D;
        const bool is_v = head >= p.H + p.Hkv;
        const bool is_k = head >= p.H && !is_v;
        const bool roped = cosb != nullptr && !is_v;
        if (roped && j >= rh8 && j < 2 * rh8) continue;  // second half: handled with its partner
        const int slot = idx == first ? pre.slot : p.slots[t];
        const int page = slot >> 5, tok = slot & 31;
        const int64_t srow = ((t >> 5) * p.S * 32 + (t & 31)) * p.slab_ld;
        const V8 a = sum_slabs_t<T>(rs, srow + head * p.D + j * 8, 32 * p.slab_ld, p.S, bias, head * p.D + j * 8);
        if (roped && j < rh8) {
            const V8 b = sum_slabs_t<T>(rs, srow + head * p.D + (j + rh8) * 8, 32 * p.slab_ld, p.S, bias,
                                        head * p.D + (j + rh8) * 8);
            V8 c = pre.c, s = pre.s;
            if (idx != first) {
                c = ld16<V8>(cosb + (int64_t)p.positions[t] * (p.rot >> 1) + j * 8);
                s = ld16<V8>(sinb + (int64_t)p.positions[t] * (p.rot >> 1) + j * 8);
            }
            V8 o1, o2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x1 = to_f32(a[e]), x2 = to_f32(b[e]), cf = to_f32(c[e]), sf = to_f32(s[e]);
                o1[e] = from_f32<T>(x1 * cf - x2 * sf);
                o2[e] = from_f32<T>(x1 * sf + x2 * cf);
            }
            st16(hp + j * 8, o1);
            st16(hp + (j + rh8) * 8, o2);
            if (is_k) {
                T* kb = kpool + ((int64_t)page * p.Hkv + (head - p.H)) * 32 * p.D;
                st16(kb + k_off(tok, j * 8, p.D), o1);
                st16(kb + k_off(tok, (j + rh8) * 8, p.D), o2);
            }
        } else if (is_k) {
            st16(hp + j * 8, a);
            T* kb = kpool + ((int64_t)page * p.Hkv + (head - p.H)) * 32 * p.D;
            st16(kb + k_off(tok, j * 8, p.D), a);
        } else if (is_v) {
            st16(hp + j * 8, a);
            // V block: [4 column groups][D][8] (csrc/kv_layout.h, round 4)
            const int cp = v_col(tok);
            T* vb = vpool + ((int64_t)page * p.Hkv + (head - p.H - p.Hkv)) * 32 * p.D + ((int64_t)((cp >> 3) * p.D + j * 8) << 3) + (cp & 7);
#pragma unroll
            for (int e = 0; e < 8; ++e) vb[e * 8] = a[e];
        } else {
            st16(hp + j * 8, a);  // q chunk outside the rotary span
        }
    }
}

template <class F, int TW0, int TW1, int TW2, int TW3>
__global__ __launch_bounds__(F::THREADS) void llama_decode_tail_kernel(TailArgsT<typename F::Args> t) {
    using T = typename F::T;
    constexpr int THREADS = F::THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 16) reinterpret_cast<int*>(smem + gptq::TAIL_CTRL)[threadIdx.x] = 0;
    BarCtx bc = bar_init(t.bar);
    __syncthreads();
    int ub = 0;  // value of the LDS unit-barrier counter, advanced identically by every wave of the workgroup
    int stamp_i = 0;
    auto stamp = [&]() {
        if (t.trace && threadIdx.x == 0) t.trace[blockIdx.x * 16 + stamp_i] = __builtin_amdgcn_s_memrealtime();
        ++stamp_i;
    };
    auto nothing = []() {};
    typename F::Ring ring;

    stamp();
    gemm_phase<F, 0, FULL, TW0>(t.g[0], t.gx[0], t.gy[0], smem, wave, ub, ring);  // o_proj -> slabs
    stamp();
    // the weights of gate_up do not depend on anything computed here: they stream through the norm phase
    NormPre<T> npre;
    grid_sync(t.bar, bc, [&]() {
        norm_prefetch<T>(t.n[0], npre);
        gemm_phase<F, F::ACT1, PF, TW1>(t.g[1], t.gx[1], t.gy[1], smem, wave, ub, ring);
    });
    stamp();
    run_norm<T, THREADS>(t.n[0], smem, npre);                                        // + residual, post-attention norm
    stamp();
    grid_sync(t.bar, bc, nothing);
    stamp();
    gemm_phase<F, F::ACT1, RUN, TW1>(t.g[1], t.gx[1], t.gy[1], smem, wave, ub, ring);  // gate_up
    stamp();
    grid_sync(t.bar, bc, [&]() { gemm_phase<F, F::ACT2, PF, TW2>(t.g[2], t.gx[2], t.gy[2], smem, wave, ub, ring); });
    stamp();
    gemm_phase<F, F::ACT2, RUN, TW2>(t.g[2], t.gx[2], t.gy[2], smem, wave, ub, ring);  // down -> slabs
    stamp();
    if (t.phases <= 5) {
        grid_sync(t.bar, bc, [&]() { norm_prefetch<T>(t.n[1], npre); });
        stamp();
        run_norm<T, THREADS>(t.n[1], smem, npre);                                    // + residual, final norm
        stamp();
        return;
    }
    grid_sync(t.bar, bc, [&]() {
        norm_prefetch<T>(t.n[1], npre);
        gemm_phase<F, 0, PF, TW3>(t.g[3], t.gx[3], t.gy[3], smem, wave, ub, ring);
    });
    stamp();
    run_norm<T, THREADS>(t.n[1], smem, npre);                                        // + residual, next layer's input norm
    stamp();
    grid_sync(t.bar, bc, nothing);
    stamp();
    gemm_phase<F, 0, RUN, TW3>(t.g[3], t.gx[3], t.gy[3], smem, wave, ub, ring);      // next layer's qkv -> slabs
    stamp();
    RopePre<T> rpre;
    grid_sync(t.bar, bc, [&]() { rope_prefetch<T, THREADS>(t.r, rpre); });
    stamp();
    run_rope<T, THREADS>(t.r, rpre);                                                 // rotary + KV pages of the next layer
    stamp();
}

// Plan signatures (TN * 10 + WK of o_proj, gate_up, down, qkv) the kernels are built for; other models keep the
// separate launches.  int4: 24 34 24 42 = Llama-2-7B (E 4096, I 11008, MHA) at M <= 32.
// dense: 22 24 22 22 = TinyLlama-1.1B (E 2048, I 5632, GQA 32:4) — plan_dense_tail.
#define TGIS_TAIL_SIGNATURES(X) X(24, 34, 24, 42)
#define TGIS_DENSE_TAIL_SIGNATURES(X) X(22, 24, 22, 22)

typedef void (*TailKernel)(TailArgs);
typedef void (*DenseTailKernel)(DenseTailArgs);
TailKernel tail_kernel_for(const int* tw) {
#define X(A, B, C, D) \
    if (tw[0] == A && tw[1] == B && tw[2] == C && (tw[3] == D || tw[3] < 0)) return llama_decode_tail_kernel<GptqFamily, A, B, C, D>;
    TGIS_TAIL_SIGNATURES(X)
#undef X
    return nullptr;
}
template <typename T>
DenseTailKernel dense_tail_kernel_for(const int* tw) {
#define X(A, B, C, D) \
    if (tw[0] == A && tw[1] == B && tw[2] == C && (tw[3] == D || tw[3] < 0)) return llama_decode_tail_kernel<DenseFamily<T>, A, B, C, D>;
    TGIS_DENSE_TAIL_SIGNATURES(X)
#undef X
    return nullptr;
}

GridBar* g_bar[16] = {};
long long* g_trace_buf[16] = {};

int fill_gemm(GemmArgs& a, int& tw, int& gx, int& gy, const tgis_tail_linear& l, const void* x, int64_t ldx, void* out,
              int64_t ldo, float* slabs, int64_t M, int act, int partial) {
    TGIS_CHECK_ARG(l.prepared && l.K > 0 && l.N > 0 && l.groups > 0 && l.K % l.groups == 0, "decode tail: bad linear");
    const int64_t gs = l.K / l.groups, spg = gs / 64;
    TGIS_CHECK_ARG(gs % 64 == 0 && (l.groups == 1 || (spg & (spg - 1)) == 0),
                   "decode tail: group size %ld is not 64 * 2^n", (long)gs);
    const gptq::PrepLayout p = gptq::prep_layout(l.K, l.N, l.groups);
    const GemmPlan pl = gptq::plan_gemm(l.K, l.N, act, M);
    TGIS_CHECK_ARG(pl.MR == 1, "decode tail: M must be <= 32");
    a.x = (const f16*)x;
    a.ldx = ldx;
    a.prep = (const uint8_t*)l.prepared;
    a.offB = p.offB;
    a.bias = partial ? nullptr : (const f16*)l.bias;
    a.perm = nullptr;
    a.out = (f16*)out;
    a.ldo = ldo;
    a.M = (int)M;
    a.K = (int)l.K;
    a.N = (int)l.N;
    a.G = (int)l.groups;
    a.gs = (int)gs;
    a.KR = pl.KR;
    a.S = pl.S;
    a.NT = (int)p.NT;
    a.KS = (int)p.KS;
    a.slabs = slabs;
    a.partial = partial;
    a.spg_shift = 30;
    a.err = nullptr;
    if (l.groups > 1)
        for (a.spg_shift = 0; (1 << a.spg_shift) < spg; ++a.spg_shift) {}
    tw = pl.TN * 10 + pl.WK;
    gx = (int)cdiv64(p.NT, pl.TN);
    gy = pl.S;
    return TGIS_OK;
}

constexpr int64_t DENSE_TAIL_UNITS = 256;  // plans are made for the MI355X's 256 resident workgroups

// direct: the unit writes the model-dtype output itself (no k-split); otherwise it leaves fp32 slabs
int fill_dense(DenseArgs& a, int& tw, int& gx, int& gy, const tgis_tail_linear& l, const void* x, int64_t ldx, void* out,
               int64_t ldo, float* slabs, int64_t M, bool direct) {
    TGIS_CHECK_ARG(l.prepared && l.K > 0 && l.N > 0 && l.groups == 0 && l.K % 8 == 0 && l.N % 8 == 0,
                   "decode tail: bad dense linear");
    const dense::DensePlan pl = dense::plan_dense_tail(l.K, l.N, direct, DENSE_TAIL_UNITS);
    a.x = x;
    a.ldx = ldx;
    a.prep = (const uint8_t*)l.prepared;
    a.bias = direct ? l.bias : nullptr;
    a.out = out;
    a.ldo = ldo;
    a.M = (int)M;
    a.K = (int)l.K;
    a.N = (int)l.N;
    a.KR = pl.KR;
    a.S = pl.S;
    a.NT = (int)cdiv64(l.N, 32);
    a.KS = (int)cdiv64(l.K, 64);
    a.out_f32 = 0;
    a.slabs = slabs;
    a.partial = direct ? 0 : 1;
    a.gelu = 0;
    a.err = nullptr;
    tw = pl.TN * 10 + pl.WK;
    gx = (int)cdiv64(a.NT, pl.TN);
    gy = pl.S;
    return TGIS_OK;
}

bool is_dense(const tgis_tail_args* t) { return t->o_proj.groups == 0; }

}  // namespace

extern "C" int64_t tgis_llama_decode_tail_slab_bytes(int64_t M, int64_t K, int64_t N, int64_t groups) {
    if (M < 1 || M > 32 || K <= 0 || N <= 0) return 0;
    const int S = groups == 0 ? dense::plan_dense_tail(K, N, false, DENSE_TAIL_UNITS).S : gptq::plan_gemm(K, N, 0, M).S;
    return (int64_t)S * 32 * cdiv64(N, 32) * 32 * 4;
}

namespace {

// The checks and the norm / rope phases both families share.  g[i] / tw / gx / gy must already be filled.
template <class TA>
int finish_tail(const tgis_tail_args* t, TA& a, int threads, bool check_buffers) {
    const bool full = t->qkv.prepared != nullptr;
    TGIS_CHECK_ARG(t->hidden > 0 && t->hidden % 8 == 0 && t->hidden <= threads * 8,
                   "tgis_llama_decode_tail: bad hidden size");
    if (check_buffers) {
        TGIS_CHECK_ARG(t->attn_out && t->residual_in && t->y1 && t->res1 && t->act && t->y2 && t->res2 && t->slabs_o &&
                           t->slabs_down && t->norm1_weight && t->norm2_weight,
                       "tgis_llama_decode_tail: null tensor");
        TGIS_CHECK_ARG(!full || (t->slabs_qkv && t->qkv_out && t->slots && t->k_pool && t->v_pool && t->H > 0 &&
                                 t->Hkv > 0 && t->D > 0 && t->D % 16 == 0 && (t->cos == nullptr) == (t->sin == nullptr) &&
                                 (!t->cos || (t->positions && t->rot_dim > 0 && t->rot_dim <= t->D && t->rot_dim % 16 == 0))),
                       "tgis_llama_decode_tail: incomplete qkv / rope arguments");
    }
    const int64_t M = t->M, E = t->hidden;
    a.n[0] = NormPhase{t->slabs_o, a.g[0].S, (int64_t)a.g[0].NT * 32, t->o_proj.bias, t->residual_in, t->norm1_weight,
                       t->y1, t->res1, (int)M, (int)E, t->eps};
    a.n[1] = NormPhase{t->slabs_down, a.g[2].S, (int64_t)a.g[2].NT * 32, t->down.bias, t->res1, t->norm2_weight, t->y2,
                       t->res2, (int)M, (int)E, t->eps};
    a.phases = 5;
    if (full) {
        TGIS_CHECK_ARG(t->qkv.K == E && (!check_buffers || t->qkv.N == (int64_t)(t->H + 2 * t->Hkv) * t->D),
                       "tgis_llama_decode_tail: qkv shape");
        a.r = RopePhase{t->slabs_qkv, a.g[3].S, (int64_t)a.g[3].NT * 32, t->qkv.bias, t->qkv_out, t->qkv.N, t->cos,
                        t->sin, t->positions, t->slots, t->k_pool, t->v_pool, (int)M, t->H, t->Hkv, t->D, t->rot_dim};
        a.phases = 7;
    } else {
        a.g[3] = a.g[2];
        a.tw[3] = -1;  // no fourth GEMM: any instantiation whose first three plans match
        a.gx[3] = a.gy[3] = 0;
        a.r = RopePhase{};
    }
    return TGIS_OK;
}

// Validates the shapes and builds the kernel arguments; `kernel` = the instantiation for the four plans, or an error.
// With check_buffers = false only M, hidden and the four linears' shapes are looked at (tgis_llama_decode_tail_fits).
int prepare_tail(const tgis_tail_args* t, TailArgs& a, TailKernel& kernel, bool check_buffers) {
    TGIS_CHECK_ARG(t->M >= 1 && t->M <= 32, "tgis_llama_decode_tail: M (%ld) must be 1..32", (long)t->M);
    TGIS_CHECK_ARG(t->dtype == TGIS_F16, "tgis_llama_decode_tail: int4 layers run in f16");
    const int64_t M = t->M, E = t->hidden;
    int rc;
    // o_proj: attn_out [M, K_o] -> slabs_o
    if ((rc = fill_gemm(a.g[0], a.tw[0], a.gx[0], a.gy[0], t->o_proj, t->attn_out, t->o_proj.K, nullptr, 0, t->slabs_o, M, 0, 1)))
        return rc;
    TGIS_CHECK_ARG(t->o_proj.N == E && t->down.N == E && t->gate_up.K == E, "tgis_llama_decode_tail: shapes do not chain");
    TGIS_CHECK_ARG(t->gate_up.N % 32 == 0 && t->down.K == t->gate_up.N / 2, "tgis_llama_decode_tail: gate_up / down mismatch");
    if ((rc = fill_gemm(a.g[1], a.tw[1], a.gx[1], a.gy[1], t->gate_up, t->y1, E, t->act, t->down.K, nullptr, M, 2, 0)))
        return rc;
    TGIS_CHECK_ARG(a.g[1].S == 1, "tgis_llama_decode_tail: gate_up must not be k-split");
    if ((rc = fill_gemm(a.g[2], a.tw[2], a.gx[2], a.gy[2], t->down, t->act, t->down.K, nullptr, 0, t->slabs_down, M, 0, 1)))
        return rc;
    if (t->qkv.prepared &&
        (rc = fill_gemm(a.g[3], a.tw[3], a.gx[3], a.gy[3], t->qkv, t->y2, E, nullptr, 0, t->slabs_qkv, M, 0, 1)))
        return rc;
    if ((rc = finish_tail(t, a, GptqFamily::THREADS, check_buffers))) return rc;
    kernel = tail_kernel_for(a.tw);
    TGIS_CHECK_ARG(kernel, "tgis_llama_decode_tail: no kernel was built for the plans %d %d %d %d", a.tw[0], a.tw[1],
                   a.tw[2], a.tw[3]);
    return TGIS_OK;
}

// Dense layers: same chain; gate_up is a tgis_dense_prepare image with flags bit 0 and is never k-split.
int prepare_dense_tail(const tgis_tail_args* t, DenseTailArgs& a, DenseTailKernel& kernel, bool check_buffers) {
    TGIS_CHECK_ARG(t->M >= 1 && t->M <= 32, "tgis_llama_decode_tail: M (%ld) must be 1..32", (long)t->M);
    TGIS_CHECK_ARG(t->dtype == TGIS_F16 || t->dtype == TGIS_BF16, "tgis_llama_decode_tail: bad dtype");
    const int64_t M = t->M, E = t->hidden;
    int rc;
    if ((rc = fill_dense(a.g[0], a.tw[0], a.gx[0], a.gy[0], t->o_proj, t->attn_out, t->o_proj.K, nullptr, 0, t->slabs_o, M, false)))
        return rc;
    TGIS_CHECK_ARG(t->o_proj.N == E && t->down.N == E && t->gate_up.K == E, "tgis_llama_decode_tail: shapes do not chain");
    TGIS_CHECK_ARG(t->gate_up.N % 32 == 0 && t->down.K == t->gate_up.N / 2, "tgis_llama_decode_tail: gate_up / down mismatch");
    if ((rc = fill_dense(a.g[1], a.tw[1], a.gx[1], a.gy[1], t->gate_up, t->y1, E, t->act, t->down.K, nullptr, M, true)))
        return rc;
    if ((rc = fill_dense(a.g[2], a.tw[2], a.gx[2], a.gy[2], t->down, t->act, t->down.K, nullptr, 0, t->slabs_down, M, false)))
        return rc;
    if (t->qkv.prepared &&
        (rc = fill_dense(a.g[3], a.tw[3], a.gx[3], a.gy[3], t->qkv, t->y2, E, nullptr, 0, t->slabs_qkv, M, false)))
        return rc;
    using F = DenseFamily<f16>;
    if ((rc = finish_tail(t, a, F::THREADS, check_buffers))) return rc;
    kernel = t->dtype == TGIS_F16 ? dense_tail_kernel_for<f16>(a.tw) : dense_tail_kernel_for<bf16>(a.tw);
    TGIS_CHECK_ARG(kernel, "tgis_llama_decode_tail: no dense kernel was built for the plans %d %d %d %d", a.tw[0],
                   a.tw[1], a.tw[2], a.tw[3]);
    return TGIS_OK;
}

int device_cus(int& dev, int& ncu) {
    TGIS_CHECK_HIP(hipGetDevice(&dev));
    TGIS_CHECK_ARG(dev >= 0 && dev < 16, "tgis_llama_decode_tail: device index");
    static int cus[16] = {};
    if (!g_bar[dev]) {
        hipDeviceProp_t prop;
        TGIS_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        GridBar* b = nullptr;
        TGIS_CHECK_HIP(hipMalloc((void**)&b, sizeof(GridBar)));
        TGIS_CHECK_HIP(hipMemset(b, 0, sizeof(GridBar)));
        g_bar[dev] = b;
        cus[dev] = prop.multiProcessorCount;
    }
    ncu = cus[dev];
    return TGIS_OK;
}

template <class TA>
int launch_tail(void (*kernel)(TA), TA& a, int threads, hipStream_t st) {
    int dev = 0, ncu = 0;
    int rc = device_cus(dev, ncu);
    if (rc != TGIS_OK) return rc;
    static std::vector<std::pair<const void*, int>> ready;  // (kernel, device) pairs whose attributes are set
    bool seen = false;
    for (auto& kd : ready) seen = seen || (kd.first == (const void*)kernel && kd.second == dev);
    if (!seen) {
        TGIS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, gptq::TAIL_LDS));
        int per_cu = 0;
        TGIS_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, gptq::TAIL_LDS));
        TGIS_CHECK_ARG(per_cu >= 1, "tgis_llama_decode_tail: the kernel does not fit a CU");
        ready.emplace_back((const void*)kernel, dev);
    }
    for (int i = 0; i < 4; ++i)  // the prefetch hands a ring from one phase to the next: one unit per workgroup and phase
        TGIS_CHECK_ARG(a.gx[i] * a.gy[i] <= ncu, "tgis_llama_decode_tail: more units than workgroups");
    a.bar = g_bar[dev];
    for (int i = 0; i < 4; ++i) a.g[i].err = &g_bar[dev]->err;
    a.trace = g_trace_buf[dev];
    TgisTimedScope timed(TGIS_OP_DECODE_TAIL, st);
    hipLaunchKernelGGL(kernel, dim3((unsigned)ncu), dim3(threads), gptq::TAIL_LDS, st, a);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

}  // namespace

// 1 if the tail can run a layer with these shapes (only M, hidden, dtype and the K / N / groups of the linears are
// read; qkv.K == 0 asks about a last layer, without the fourth GEMM)
extern "C" int tgis_llama_decode_tail_fits(const tgis_tail_args* t) {
    if (!t) return 0;
    tgis_tail_args c = *t;
    static const char dummy = 0;  // shapes only: any non-null image pointer
    c.o_proj.prepared = c.gate_up.prepared = c.down.prepared = &dummy;
    c.qkv.prepared = c.qkv.K > 0 ? &dummy : nullptr;
    int gx[4], gy[4], rc;
    if (is_dense(&c)) {
        DenseTailArgs a;
        DenseTailKernel k = nullptr;
        rc = prepare_dense_tail(&c, a, k, false);
        for (int i = 0; i < 4; ++i) gx[i] = a.gx[i], gy[i] = a.gy[i];
    } else {
        TailArgs a;
        TailKernel k = nullptr;
        rc = prepare_tail(&c, a, k, false);
        for (int i = 0; i < 4; ++i) gx[i] = a.gx[i], gy[i] = a.gy[i];
    }
    tgis_clear_error();
    if (rc != TGIS_OK) return 0;
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) ncu = 256;  // no device (build machine): the MI355X count
    for (int i = 0; i < 4; ++i)
        if (gx[i] * gy[i] > ncu) return 0;  // one unit per workgroup and phase
    return 1;
}

extern "C" int tgis_llama_decode_tail(const tgis_tail_args* t, void* stream) {
    TGIS_CHECK_ARG(t, "tgis_llama_decode_tail: null arguments");
    hipStream_t st = (hipStream_t)stream;
    if (is_dense(t)) {
        DenseTailArgs a;
        DenseTailKernel kernel = nullptr;
        int rc = prepare_dense_tail(t, a, kernel, true);
        if (rc != TGIS_OK) return rc;
        return launch_tail(kernel, a, DenseFamily<f16>::THREADS, st);
    }
    TailArgs a;
    TailKernel kernel = nullptr;
    int rc = prepare_tail(t, a, kernel, true);
    if (rc != TGIS_OK) return rc;
    return launch_tail(kernel, a, GptqFamily::THREADS, st);
}

// Debug: from the next launch on every workgroup records s_memrealtime (100 MHz) at the edges of its phases; `out`
// receives [workgroups][16] stamps of the LAST launch (call after synchronising).  enable < 0 only reads.
extern "C" int tgis_llama_decode_tail_trace(int enable, long long* out, int max_workgroups) {
    int dev = 0;
    TGIS_CHECK_HIP(hipGetDevice(&dev));
    TGIS_CHECK_ARG(dev >= 0 && dev < 16, "tgis_llama_decode_tail_trace: device index");
    if (enable > 0 && !g_trace_buf[dev]) {
        TGIS_CHECK_HIP(hipMalloc((void**)&g_trace_buf[dev], 1024 * 16 * sizeof(long long)));
        TGIS_CHECK_HIP(hipMemset(g_trace_buf[dev], 0, 1024 * 16 * sizeof(long long)));
    }
    if (out && g_trace_buf[dev])
        TGIS_CHECK_HIP(hipMemcpy(out, g_trace_buf[dev], (size_t)std::min(max_workgroups, 1024) * 16 * sizeof(long long),
                                 hipMemcpyDeviceToHost));
    if (enable == 0 && g_trace_buf[dev]) {
        (void)hipFree(g_trace_buf[dev]);
        g_trace_buf[dev] = nullptr;
    }
    return TGIS_OK;
}

// 0 = no barrier of this device ever timed out; resets the flag and the barrier state when it did (debug / recovery).
extern "C" int tgis_llama_decode_tail_status(int reset) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || !g_bar[dev]) return 0;
    unsigned err = 0;
    if (hipMemcpy(&err, &g_bar[dev]->err, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (err && reset) (void)hipMemset(g_bar[dev], 0, sizeof(GridBar));
    return (int)err;
}
