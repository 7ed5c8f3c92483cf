// Pieces for a SECOND phase inside one launch (round 3): a fence-free grid barrier over workgroups that are all resident,
// L1-bypassing 16-byte accesses for data that crosses it, and the add + RMSNorm of norm.hip as a device function that a
// GEMM kernel runs as its first phase.  Same protocol as decode_tail.hip (which keeps its own copies): inter-workgroup
// data travels as sc1 (write-through) stores, drained with vmcnt(0) before the workgroup arrives, and sc1 loads after
// the barrier; the barrier itself is relaxed agent-scope atomics only (MI355X_MICROARCH.md, inter-workgroup visibility;
// 1.8 us at 256 workgroups, tools/floor/xcdbar.hip).  Every spin is bounded: a barrier that cannot complete (a workgroup
// not resident) leaves a code in GridBar::err and the launch goes on with garbage instead of hanging the device.
#pragma once
#include "common.h"

namespace gsync {

constexpr unsigned SPIN_LIMIT = 1u << 22;

// One 128-byte line per hot word.  The arrival counters (`grp`, `top`) are reset by their last arriver, so consecutive
// launches may have different grid sizes (decode_tail.hip's copy counts monotonically: its grid never changes); only the
// generation words are monotonic, all eight always move together, and a launch reads the generation it starts from.
struct GridBar {
    unsigned top, pad0[31];
    unsigned grp[8][32];
    unsigned gen[8][32];
    unsigned err, pad1[31];
};

struct BarCtx {
    unsigned epoch;   // generation of the barrier this workgroup arrives at next (thread 0)
    unsigned gsize;   // workgroups in this workgroup's group
    unsigned ngroups;
};

#define GS_RLX_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GS_RLX_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GS_RLX_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// one-dimensional grids only (blockIdx.x is the workgroup id)
__device__ __forceinline__ BarCtx bar_init(GridBar* b) {
    BarCtx c;
    const unsigned g = blockIdx.x & 7u;
    c.ngroups = min(8u, gridDim.x);
    c.gsize = (gridDim.x - g + 7u) / 8u;
    c.epoch = 0;
    if (threadIdx.x == 0) c.epoch = GS_RLX_LOAD(&b->gen[g][0]);  // all workgroups of a launch read the same value
    return c;
}

// `between` runs on every wave after the workgroup has arrived and before it waits: loads issued there (the weights of
// the phase behind the barrier) are in flight while the barrier completes.
template <class F>
__device__ __forceinline__ void grid_sync(GridBar* b, BarCtx& c, F&& between) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have left the CU
    __syncthreads();
    const unsigned e = c.epoch + 1u;
    const unsigned g = blockIdx.x & 7u;
    if (threadIdx.x == 0) {
        c.epoch = e;
        const unsigned old = GS_RLX_ADD(&b->grp[g][0], 1u);
        if (old + 1u == c.gsize) {  // last of the group: nobody touches the counter again before the next launch
            GS_RLX_STORE(&b->grp[g][0], 0u);
            const unsigned t = GS_RLX_ADD(&b->top, 1u);
            if (t + 1u == c.ngroups) {
                GS_RLX_STORE(&b->top, 0u);
                for (unsigned gg = 0; gg < 8u; ++gg) GS_RLX_STORE(&b->gen[gg][0], e);
            }
        }
    }
    between();
    if (threadIdx.x == 0) {
        for (unsigned spins = 0; (int)(GS_RLX_LOAD(&b->gen[g][0]) - e) < 0; ++spins) {
            __builtin_amdgcn_s_sleep(1);
            if (spins > SPIN_LIMIT) {
                GS_RLX_STORE(&b->err, 1u);
                break;
            }
        }
    }
    // execution barrier only: __syncthreads() would also wait (vmcnt) for the loads `between` has just issued
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ u32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, int64_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)byte_off, 0, 16);
}
__device__ __forceinline__ void st_sc1(u32x4 v, __amdgpu_buffer_rsrc_t r, int64_t byte_off) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (uint32_t)byte_off, 0, 16);
}

// The add + RMSNorm in front of a GEMM (LlamaRMSNorm.forward, flash_llama_modeling.py:132-152) as that GEMM's first phase.
// x either as a model-dtype tensor, or as S fp32 split-K slabs [row / 32][S][32][slab_ld] of the GEMM before (+ xbias),
// summed in slab order and rounded to T first — the arithmetic of norm.hip, bit for bit.
struct NormPhase {
    const float* slabs;   // or nullptr
    int S;
    int64_t slab_ld;
    const void* xbias;    // bias of the producing GEMM, or nullptr (partial input only)
    const void* x;        // T [rows, hidden] when slabs == nullptr
    const void* residual; // T [rows, hidden] or nullptr
    const void* weight;   // T [hidden]
    void* y;              // T [rows, hidden]: the normed activation = the GEMM's operand (written sc1)
    void* res_out;        // T [rows, hidden]: x (+ residual), the residual stream (may alias nothing the launch reads)
    int rows, hidden;
    float eps;
    int y_frag;           // 1: y leaves in 32-row fragment order (xf_off in common.h; rows <= 64, hidden % 64 == 0)
};

// One row by all `nthreads` threads of the workgroup (nthreads a multiple of 64, <= 1024); `sh` = >= 16 floats of LDS that
// nothing else uses during the phase.  MAXV bounds hidden: hidden <= MAXV * nthreads * 8.
// SC1: outputs as write-through stores (another workgroup of the same launch reads them) or plain stores (norm.hip's own
// kernel runs this very function, so that the two paths cannot drift apart by a rounding).
template <typename T, int MAXV, bool SC1 = true>
__device__ __forceinline__ void norm_row(const NormPhase& p, const int row, float* sh, const int nthreads) {
    using V8 = typename VecT<T>::x8;
    const int nchunk = p.hidden >> 3;
    const T* xr = p.slabs ? nullptr : reinterpret_cast<const T*>(p.x) + (int64_t)row * p.hidden;
    const T* rr = p.residual ? reinterpret_cast<const T*>(p.residual) + (int64_t)row * p.hidden : nullptr;
    const T* xb = reinterpret_cast<const T*>(p.xbias);
    __amdgpu_buffer_rsrc_t ry = rsrc_of(p.y), ro = rsrc_of(p.res_out);
    float v[MAXV][8];
    V8 wv[MAXV];  // the norm weights: asked for with the inputs, not after the row reduction (a second round trip)
    float s2 = 0.f;
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
        const int c = threadIdx.x + it * nthreads;
        if (c < nchunk) {
            wv[it] = ld16<V8>(reinterpret_cast<const T*>(p.weight) + c * 8);
            // Round 5: the residual and the bias of the GEMM before are requested HERE, in front of the slabs.  Behind them (in
            // their own basic blocks, after the waits of the slab sum) they were a second and a third dependent round trip to
            // memory in a kernel that is nothing but one round trip (ISA of norm_kernel<f16, true, true, 512>).
            V8 b, bv;
            if (rr) b = ld16<V8>(rr + c * 8);
            if (p.slabs && xb) bv = ld16<V8>(xb + c * 8);
            V8 a;
            if (p.slabs) {
                f32x4 lo, hi;
                sum_slabs8(p.slabs + ((int64_t)(row >> 5) * p.S * 32 + (row & 31)) * p.slab_ld + c * 8, 32 * p.slab_ld, p.S,
                           lo, hi);
                if (xb) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] += to_f32(bv[e]);
                        hi[e] += to_f32(bv[e + 4]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = from_f32<T>(lo[e]);
                    a[e + 4] = from_f32<T>(hi[e]);
                }
            } else {
                a = ld16<V8>(xr + c * 8);
            }
            if (rr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = to_f32(a[e]) + to_f32(b[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = to_f32(a[e]);
            }
            V8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = from_f32<T>(v[it][e]);
                s2 += v[it][e] * v[it][e];
            }
            if (p.res_out) {
                if (SC1)
                    st_sc1(__builtin_bit_cast(u32x4, o), ro, ((int64_t)row * p.hidden + c * 8) * 2);
                else
                    st16(reinterpret_cast<T*>(p.res_out) + (int64_t)row * p.hidden + c * 8, o);
            }
        }
    }
    // block sum of s2 in wave order (as norm.hip's block_sum)
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = s2;
    __syncthreads();
    float tot = 0.f;
    for (int k = 0; k < (nthreads >> 6); ++k) tot += sh[k];
    const float rstd = rsqrtf(tot / p.hidden + p.eps);
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
        const int c = threadIdx.x + it * nthreads;
        if (c < nchunk) {
            V8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = from_f32<T>((v[it][e] - 0.f) * rstd * to_f32(wv[it][e]));
            if (SC1)
                st_sc1(__builtin_bit_cast(u32x4, o), ry, ((int64_t)row * p.hidden + c * 8) * 2);
            else
                st16(reinterpret_cast<T*>(p.y) + (p.y_frag ? xf_off(row, c * 8, p.hidden) : (int64_t)row * p.hidden + c * 8), o);
        }
    }
}

}  // namespace gsync
