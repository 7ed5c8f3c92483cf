// GPTQ int4 linear for gfx950: one-time repack ("prepare"), fused dequant + MFMA GEMM for decode-sized
// M, and a full dequant kernel for the large-M (library GEMM) path.
//
// Replaces exllamav2_kernels.make_q_matrix / gemm_half_q_half as called from
// utils/gptq/exllamav2.py:14-62,124-144.  Normative arithmetic (utils/gptq/quant_linear.py:130-138,
// 184-194):  W[k,n] = (q[k,n] - (z[g(k),n] + 1)) * s[g(k),n];  y = x @ W, fp32 accumulate, f16 out.
//
// Prepared image, NT = ceil(N/32) column tiles, KS = ceil(K/64) k-steps, G groups:
//   A: wq [NT][KS][64 lanes][4] int32 — lane l word i = the 8 nibbles of rows
//        k = (ks*8 + (l>>5)*4 + i)*8 + {0,2,4,6,1,3,5,7} of column n = nt*32 + (l&31): one KiB per wave
//        load, and (after dequant8) exactly the B-operand fragment of v_mfma_f32_32x32x16_f16
//   B: sz [NT][G][32] u32 = { scale f16 , (1024 + z + 1) f16 }
// Rows are pre-permuted by the act-order permutation when g_idx is not trivial.
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include "common.h"
#include <type_traits>

namespace {

struct PrepLayout {
    int64_t NT, KS, G, offB, total;
};
static PrepLayout prep_layout(int64_t K, int64_t N, int64_t G) {
    PrepLayout p;
    p.NT = cdiv64(N, 32);
    // whole 256-row chunks (rows >= K hold zero nibbles; x is zero there) + one pad step so that the tile
    // stride is not a multiple of 64 KiB: equal-phase waves would otherwise camp on the same HBM channels
    p.KS = cdiv64(K, 256) * 4 + 1;
    p.G = G;
    p.offB = p.NT * p.KS * 1024;
    p.total = p.offB + p.NT * G * 128;
    p.total = (p.total + 255) & ~int64_t(255);
    return p;
}

// Column held by lane c (0..31) of tile nt.  flags bit 0 (gate/up interleave, for the fused SiLU*mul epilogue):
// lanes 0-15 hold gate columns 16 nt + c, lanes 16-31 the matching up columns N/2 + 16 nt + (c - 16).
__device__ __forceinline__ int64_t col_src(int64_t nt, int c, int64_t N, int flags) {
    if (flags & 1) return (c < 16) ? nt * 16 + c : (N >> 1) + nt * 16 + (c - 16);
    return nt * 32 + c;
}

__device__ __forceinline__ int nib_src(int j) {  // stored nibble j holds row offset {0,2,4,6,1,3,5,7}[j]
    return (j < 4) ? 2 * j : 2 * (j - 4) + 1;
}

__global__ void gptq_prepare_w_kernel(const int32_t* __restrict__ qweight, const int32_t* __restrict__ perm,
                                      int32_t* __restrict__ wq, int64_t K, int64_t N, int64_t NT, int64_t KS,
                                      int flags) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = NT * KS * 256;
    if (idx >= total) return;
    int i = idx & 3;
    int l = (idx >> 2) & 63;
    int64_t ks = (idx >> 8) % KS;
    int64_t nt = (idx >> 8) / KS;
    int64_t n = col_src(nt, l & 31, N, flags);
    int64_t p = ks * 8 + (l >> 5) * 4 + i;  // k-pack row (8 k each)
    uint32_t v = 0;
    if (n < N && p * 8 < K) {
        if (perm == nullptr) {
            uint32_t w = (uint32_t)qweight[p * N + n];
#pragma unroll
            for (int j = 0; j < 8; ++j) v |= ((w >> (4 * nib_src(j))) & 15u) << (4 * j);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int64_t ksrc = perm[p * 8 + nib_src(j)];
                uint32_t w = (uint32_t)qweight[(ksrc >> 3) * N + n];
                v |= ((w >> (4 * (ksrc & 7))) & 15u) << (4 * j);
            }
        }
    }
    wq[idx] = (int32_t)v;
}

__global__ void gptq_prepare_sz_kernel(const int32_t* __restrict__ qzeros, const f16* __restrict__ scales,
                                       uint32_t* __restrict__ sz, int64_t N, int64_t NT, int64_t G, int flags) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NT * G * 32) return;
    int c = idx & 31;
    int64_t g = (idx >> 5) % G;
    int64_t nt = (idx >> 5) / G;
    int64_t n = col_src(nt, c, N, flags);
    f16x2 v = {(f16)0.f, (f16)1025.f};
    if (n < N) {
        uint32_t w = (uint32_t)qzeros[g * (N / 8) + (n >> 3)];
        v[0] = scales[g * N + n];
        v[1] = (f16)(float)(1024u + ((w >> (4 * (n & 7))) & 15u) + 1u);
    }
    sz[idx] = __builtin_bit_cast(uint32_t, v);
}

// 8 nibbles of q -> 8 halves (q_j - zp1) * s.  The prepared image stores the nibbles of rows k0..k7 in the
// order [k0,k2,k4,k6,k1,k3,k5,k7], so the four (low,high) pairs come out as (k0,k1),(k2,k3),(k4,k5),(k6,k7):
// the natural k order of the MFMA B fragment.  (q - zp1) is exact integer arithmetic in f16 (|values| < 2048);
// the product with the f16 scale is rounded once to f16, as in exllamav2's dequantisation.
__device__ __forceinline__ uint32_t and_or(uint32_t q, uint32_t mask, uint32_t ex) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(q), "s"(mask), "v"(ex));
    return r;
}
__device__ __forceinline__ f16x8 dequant8(uint32_t q, f16x2 zc, f16x2 zd, f16x2 sc, uint32_t EX, uint32_t M0,
                                          uint32_t M1) {
    // EX = 0x64006400 (1024.0h pair) lives in a VGPR and the masks in SGPRs: (q & mask) | EX is one VALU op
    const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
    uint32_t q2 = q >> 8;
    uint32_t a0 = and_or(q, M0, EX);   // 1024 + n0 , 1024 + n4
    uint32_t a1 = and_or(q, M1, EX);   // 1024 + 16 n1 , 1024 + 16 n5
    uint32_t a2 = and_or(q2, M0, EX);  // n2, n6
    uint32_t a3 = and_or(q2, M1, EX);  // n3, n7
    f16x2 h0 = (__builtin_bit_cast(f16x2, a0) - zc) * sc;
    f16x2 h1 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a1), r16, zd) * sc;
    f16x2 h2 = (__builtin_bit_cast(f16x2, a2) - zc) * sc;
    f16x2 h3 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a3), r16, zd) * sc;
    u32x4 packed = {__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1),
                    __builtin_bit_cast(uint32_t, h2), __builtin_bit_cast(uint32_t, h3)};
    return __builtin_bit_cast(f16x8, packed);
}

struct GemmArgs {
    const f16* x;
    int64_t ldx;
    const uint8_t* prep;
    int64_t offB;
    const f16* bias;
    const int32_t* perm;
    f16* out;
    int64_t ldo;
    int M, K, N;   // M = all rows (grid.z walks 32-row slabs)
    int G, gs;     // groups, group size
    int KR;        // k-range per block (multiple of 256)
    int S;         // global k splits
    int NT, KS;
    float* slabs;  // [Mslabs][S][32][NT*32] f32 partial sums (S > 1 or partial mode)
    int partial;   // 1: always leave fp32 slabs (deferred reduce), never write `out`
    int spg_shift; // GROUP64: log2(k64-steps per group) (30 when there is a single group)
};

#ifdef TGIS_TRACE
__device__ long long* g_trace = nullptr;  // [blocks][16 waves][32 stamps] of s_memtime (debug builds only)
#define TRACE(i)                                                                                            \
    do {                                                                                                    \
        if (g_trace) /* every lane stores the same stamp: no divergent branch, the SGPR pins stay legal */  \
            g_trace[((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 +       \
                     __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define TRACE_RT(i)                                                                                         \
    do {                                                                                                    \
        if (g_trace)                                                                                        \
            g_trace[((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 +       \
                     __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define TRACE(i)
#define TRACE_RT(i)
#endif

#define GLOBAL_AS __attribute__((address_space(1)))  // asm-pinned pointers lose their address space: restate it
#define PIN_SGPR(p) asm volatile("" : "+s"(p))

constexpr int KC = 256;     // k per LDS chunk (4 k64-steps)
constexpr int RS = KC + 8;  // LDS row stride in halves (+16 B -> conflict-free ds_read_b128)

// Streaming kernel: a block of TN*WK waves owns 32*TN columns x [k0,k1) of W.  Wave w works on column tile w % TN and
// k-part w / TN: it streams its own 32-column tile over its own contiguous KR/WK rows (1 KiB per load, one chunk = 4
// loads in flight, each slot refilled in place right after it is consumed) while every k-part group of TN waves
// double-buffers its 32 x 256 chunks of x through its own LDS region and paces itself with an LDS arrival counter (the
// block-wide barrier is only used to publish the zeroed counters and before the final reduction).  The WK partial
// accumulators are summed through LDS at the end (fixed order), so in-block k-parts add waves per SIMD without slab
// traffic.  The main loop is branch-free (clamped addresses, zero scales past the wave's rows) so that hipcc keeps
// counted vmcnt waits; everything wave-uniform lives in SGPRs.
// GROUP64: group size is 64 * 2^n (one scale/zero per lane per step, prefetched with the weights).
// TN = column tiles (waves) per k-part: 2, 3 or 4; WK = k-parts per block: 2 or 4 (plan_gemm).
// MR = 32-row blocks of x per pass: 1 (M <= 32), or 2 for larger decode batches — the wave dequantises each fragment
// once and feeds it to two MFMAs, instead of streaming and dequantising the weights again for rows 32..63.  MR = 2 needs
// WK = 2 (the x chunk buffers double).
template <int TN, int WK, int ACT, bool GROUP64, bool PERM, int MR>
__global__ __launch_bounds__(64 * TN * WK) void gptq_gemm_kernel(GemmArgs a) {
    static_assert(MR == 1 || WK == 2, "64-row passes need the LDS of two k-parts");
    constexpr int XR = 32 * MR;                 // x rows per pass
    // one chunk (4 one-KiB loads) of weights in flight per wave.  Measured: a two-chunk ring is ~1 us SLOWER on every
    // cfg3 shape (the first barrier waits for twice the prologue loads to issue; HBM is not the limiter afterwards)
    constexpr int RING = 4;
    constexpr int GT = 64 * TN;                 // threads of one k-part group
    constexpr int NJ = (XR * 32 + GT - 1) / GT;  // 16-byte x pieces per thread per chunk (XR rows x 32 pieces per chunk)
    constexpr int RSTEP = GT / 32;              // rows covered by one pass of the group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index and everything derived from it (tile, k-range, image addresses) is wave-uniform: keep it in
    // SGPRs so that the address/clamp arithmetic of the loop runs on the scalar unit, not on the VALU that the
    // dequantisation saturates
    TRACE(0);
    TRACE_RT(14);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w % TN, wk = w / TN, ltid = wn * 64 + lane;
    f16* xs = reinterpret_cast<f16*>(smem) + wk * (2 * XR * RS);  // this k-part's [2][XR][RS]
    const int ntg = blockIdx.x, split = blockIdx.y, mslab = blockIdx.z;
    const int m0 = mslab * XR;
    const int mrows = min(XR, a.M - m0);
    const int krp = a.KR / WK;                           // rows per k-part (multiple of 256)
    const int k0 = split * a.KR + wk * krp;
    const int k1 = min(a.K, k0 + krp);                   // may be <= k0 for trailing k-parts: they add zeros
    const int nchunks = krp / KC;                        // block-uniform trip count (same barriers for every wave)
    const int nt_raw = ntg * TN + wn;
    const int nt = min(nt_raw, a.NT - 1);                // out-of-range waves recompute the last tile, never store
    const int ks0 = k0 >> 6;
    const int ks_last = a.KS - 2;                        // last real step of the image (KS includes one pad step)

    const char* wtile = reinterpret_cast<const char*>(a.prep) + (int64_t)nt * a.KS * 1024;
    const char* sztile = reinterpret_cast<const char*>(a.prep + a.offB) + (int64_t)nt * a.G * 128;
    const uint32_t woff = lane * 16, szoff = (lane & 31) * 4;
    // prefetches past this wave's rows re-read its own last step (a cache hit), not the next k-part's rows
    const int ks_clamp = min(ks_last, max(ks0, ((k1 + 63) >> 6) - 1));  // K % 64 == 32: the last step is half valid
    const int ks_end = k1 >> 6;  // first step past this wave's rows: its scale is forced to zero (GROUP64: K % 64 == 0)
    auto sz_at = [&](int step) -> uint32_t {   // GROUP64: one {scale, zero} pair per lane and k64-step
        const int g = min((ks0 + step) >> a.spg_shift, a.G - 1);
        const char* p = sztile + (int64_t)g * 128;
        PIN_SGPR(p);  // keep the wave-uniform base in SGPRs: the load takes (sgpr base + lane offset)
        const uint32_t v = *(const GLOBAL_AS uint32_t*)(p + szoff);
        return ks0 + step < ks_end ? v : 0u;   // zero scale: x columns past k1 are never masked, their weights are
    };
    auto w_at = [&](int step) -> u32x4 {
        const char* p = wtile + (int64_t)min(ks0 + step, ks_clamp) * 1024;
        PIN_SGPR(p);
        return __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(p + woff));
    };
    const uint32_t* szp = reinterpret_cast<const uint32_t*>(sztile + szoff);

    // ---- x staging: local thread t handles rows (t / 32) + RSTEP j, 16-byte column piece (t & 31) ----
    // Rows past M and columns past k1 are loaded from clamped (valid, finite) addresses and never masked: a row only
    // feeds its own output row, and the weights of steps past k1 carry a zero scale.
    const f16* xbase = a.x + (int64_t)m0 * a.ldx;
    const int srow = ltid >> 5, scol = (ltid & 31) * 8;
    f16x8 xg[NJ], xu[NJ];
    uint32_t rowoff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rowoff[j] = (uint32_t)(min(srow + RSTEP * j, mrows - 1) * (int)a.ldx * 2);
    auto stage_load = [&](int chunk) {
        const int kc = min(k0 + chunk * KC + scol, a.K - 8);
        const char* xb = reinterpret_cast<const char*>(xbase);
        PIN_SGPR(xb);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f16x8 v, u;
            if (!PERM) {
                const uint32_t off = rowoff[j] + (uint32_t)kc * 2;
                v = *(const GLOBAL_AS f16x8*)(xb + off);
                if (ACT == 1) u = *(const GLOBAL_AS f16x8*)(xb + (int64_t)a.K * 2 + off);
            } else {
                const GLOBAL_AS f16* xr = (const GLOBAL_AS f16*)(xb + rowoff[j]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int ksrc = a.perm[kc + e];
                    v[e] = xr[ksrc];
                    if (ACT == 1) u[e] = xr[a.K + ksrc];
                }
            }
            xg[j] = v;
            if (ACT == 1) xu[j] = u;
        }
    };
    auto stage_store = [&](int buf) {
        f16* dst = xs + buf * (XR * RS) + srow * RS + scol;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f16x8 t = xg[j];
            if (ACT == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float gte = (float)t[e];
                    float sl = gte / (1.f + __expf(-gte));
                    // reference rounds silu(gate) to f16 before the multiply (eager torch ops)
                    t[e] = (f16)((float)(f16)sl * (float)xu[j][e]);
                }
            }
            if (NJ * RSTEP == XR || srow + RSTEP * j < XR) st16(dst + j * RSTEP * RS, t);
        }
    };

    uint32_t EXr = 0x64006400u, M0r = 0x000F000Fu, M1r = 0x00F000F0u;
    asm volatile("" : "+v"(EXr));
    asm volatile("" : "+s"(M0r), "+s"(M1r));
    // two accumulators: consecutive MFMAs of a step alternate, halving the dependent-accumulator stalls
    f32x16 accs[MR][2];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int i = 0; i < 2; ++i) accs[mr][i] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int xoff = (lane & 31) * RS + (lane >> 5) * 32;

    TRACE(1);
    // Per-k-part arrival counters (monotonic): only the TN waves that share an x buffer synchronise per chunk.  A
    // block-wide s_barrier would park every wave until the slowest of all TN*WK has finished its chunk (the SIMD
    // arbiter serves its oldest wave first, so a third of the loop time went to that skew).
    typedef __attribute__((address_space(3))) int lds_int;  // explicit LDS pointer: a generic one costs vmcnt(0) waits
    volatile lds_int* sync_cnt = (volatile lds_int*)(smem + (size_t)WK * 2 * XR * RS * sizeof(f16)) + wk;
    if (wn == 0 && lane == 0) *sync_cnt = 0;
    // Issue order matters: a wave's loads return in order, so the (L2-resident) first x chunk goes out before the
    // HBM weight stream it would otherwise queue behind; then the small scale loads, then the ring of weights.
    stage_load(0);
    u32x4 wq[RING];
    uint32_t szr[RING];
#pragma unroll
    for (int s = 0; s < RING; ++s)
        if (GROUP64) szr[s] = sz_at(s);
#pragma unroll
    for (int s = 0; s < RING; ++s) wq[s] = w_at(s);
    // the only block-wide barrier before the reduction publishes the zeroed counters; it does not wait for the loads
    // above, and from here on each k-part group paces itself
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto group_sync = [&](int target) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lds_int*)sync_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifndef ABL_NOSYNCWAIT  // ablation: arrive but never wait (wrong results; bounds what a slacker hand-off could gain)
        while (__builtin_amdgcn_readfirstlane(*sync_cnt) < target) __builtin_amdgcn_s_sleep(1);
#endif
        asm volatile("" ::: "memory");
    };
    stage_store(0);
    group_sync(TN);
    TRACE(2);

    // One chunk = 4 k64-steps.  All but the last chunk prefetch: next chunk's x into registers, next chunk's
    // scales, and each weight slot is refilled in place right after it is consumed.  The last chunk is a separate
    // instantiation without any of that (no wasted re-reads, no sync).
    auto chunk_body = [&](const int chunk, auto last_tag) {
        constexpr bool STAGE = !decltype(last_tag)::value, REFILL = STAGE;
        constexpr int SB = 0;
#ifndef ABL_NOSTAGE
        if (STAGE) stage_load(chunk + 1);
#endif
        // next chunk's scales: issued before this chunk's weight refills so that the loop-carried copy at the
        // bottom only needs vmcnt(#weight loads) and the weight stream stays in flight across the sync
        uint32_t szn[4];
        if (GROUP64 && REFILL) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) szn[s4] = sz_at(chunk * 4 + s4 + RING);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f16* xbuf = xs + (chunk & 1) * (XR * RS) + xoff;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int step = chunk * 4 + s4;
            const u32x4 cur = wq[SB + s4];
            const f16* xk = xbuf + s4 * 64;
            f16x8 b[4];
            if (GROUP64) {
                const f16x2 szh = __builtin_bit_cast(f16x2, szr[SB + s4]);
                const f16 zc1 = szh[1];
                const f16 zd1 = (f16)960.f - zc1;  // -(64 + z + 1), exact
                const f16x2 zc = {zc1, zc1}, zd = {zd1, zd1}, sc = {szh[0], szh[0]};
#ifdef ABL_NODEQ
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(f16x8, u32x4{cur[i], cur[i] ^ EXr, cur[i], cur[i]});
#else
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = dequant8(cur[i], zc, zd, sc, EXr, M0r, M1r);
#endif
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int kreal = ((ks0 + step) * 8 + (lane >> 5) * 4 + i) * 8;
                    int k = min(kreal, a.K - 8);
                    int g = min(k / a.gs, a.G - 1);
                    f16x2 szh = __builtin_bit_cast(f16x2, kreal < k1 ? szp[g * 32] : 0u);
                    f16 zc1 = szh[1], zd1 = (f16)960.f - zc1;
                    f16x2 zc = {zc1, zc1}, zd = {zd1, zd1}, sc = {szh[0], szh[0]};
                    b[i] = dequant8(cur[i], zc, zd, sc, EXr, M0r, M1r);
                }
            }
            if (REFILL) {
                // the slot is consumed: refill it in place, RC chunks ahead (no register copy at the back-edge)
                __builtin_amdgcn_sched_barrier(0);
                wq[SB + s4] = w_at(step + RING);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#if defined(ABL_NOMFMA)
                accs[0][i & 1][0] += (float)b[i][0] + (float)b[i][7];
#elif defined(ABL_NOLDSREAD)
                accs[0][i & 1] = mfma32(b[(i + 1) & 3], b[i], accs[0][i & 1]);
#else
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    f16x8 av = ld16<f16x8>(xk + mr * (32 * RS) + i * 8);
                    accs[mr][i & 1] = mfma32(av, b[i], accs[mr][i & 1]);
                }
#endif
            }
        }
        if (GROUP64 && REFILL) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) szr[SB + s4] = szn[s4];
        }
        if (!STAGE) return;
#ifndef ABL_NOSTAGE
        stage_store((chunk + 1) & 1);
#endif
        TRACE(3 + 2 * min(chunk, 3));
        group_sync(TN * (chunk + 2));
        TRACE(4 + 2 * min(chunk, 3));
    };
    for (int chunk = 0; chunk + 1 < nchunks; ++chunk) chunk_body(chunk, std::false_type{});
    chunk_body(nchunks - 1, std::true_type{});
    TRACE(9);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every k-part is done with its x buffers: the reduction below reuses them

    f32x16 acc[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) acc[mr] = accs[mr][0] + accs[mr][1];
    // ---- sum the WK k-parts through LDS (fixed order => deterministic) --------------------------------
    if (WK > 1) {
        float* red = reinterpret_cast<float*>(smem);  // [WK][TN tiles][MR][64 lanes][16]; the x buffers are dead now
        if (wk > 0) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                float* dst = red + ((((wk * TN + wn) * MR + mr) * 64 + lane) << 4);
#pragma unroll
                for (int r = 0; r < 16; r += 4)
                    *reinterpret_cast<f32x4*>(dst + r) = f32x4{acc[mr][r], acc[mr][r + 1], acc[mr][r + 2], acc[mr][r + 3]};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        TRACE(11);
        if (wk > 0) return;
#pragma unroll
        for (int k2 = 1; k2 < WK; ++k2)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                const float* src = red + ((((k2 * TN + wn) * MR + mr) * 64 + lane) << 4);
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    f32x4 t = *reinterpret_cast<const f32x4*>(src + r);
                    acc[mr][r] += t[0];
                    acc[mr][r + 1] += t[1];
                    acc[mr][r + 2] += t[2];
                    acc[mr][r + 3] += t[3];
                }
            }
    }

    // ---- epilogue: lane holds out[m = 32 mr + (r&3)+8(r>>2)+4(lane>>5)][n = nt*32 + (lane&31)] -------
    TRACE(12);
    TRACE_RT(15);
    if (nt_raw >= a.NT) return;
    const int n = nt * 32 + (lane & 31);
    if (ACT == 2) {
        // columns are interleaved gate/up pairs (col_src flags bit 0): lanes c < 16 hold gate column j = 16 nt + c,
        // lanes c + 16 the matching up column.  out[m][j] = f16(f16(silu(f16 gate)) * f16 up), the rounding
        // sequence of the reference's eager ops (flash_llama_modeling.py:332-335).  Host guarantees S == 1.
        const int c = lane & 31;
        const int half = a.N >> 1;
        const int j = nt * 16 + (c & 15);
        const int nsrc = (c < 16) ? j : half + j;
        const float bv = (a.bias && j < half) ? (float)a.bias[nsrc] : 0.f;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float mine = (float)(f16)(acc[mr][r] + bv);
                const float other = __shfl_xor(mine, 16, 64);
                const int m = mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (c < 16 && j < half && m < mrows) {
                    float sl = mine / (1.f + __expf(-mine));
                    a.out[(int64_t)(m0 + m) * a.ldo + j] = (f16)((float)(f16)sl * other);
                }
            }
        return;
    }
    if (a.S == 1 && !a.partial) {
        const float bv = (a.bias && n < a.N) ? (float)a.bias[n] : 0.f;
        if (n < a.N) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < mrows) a.out[(int64_t)(m0 + m) * a.ldo + n] = (f16)(acc[mr][r] + bv);
                }
        }
    } else {
        // slabs are indexed in 32-row units: this pass owns units mslab*MR .. mslab*MR + MR-1
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            float* sl = a.slabs + ((int64_t)((mslab * MR + mr) * a.S + split) * 32) * (a.NT * 32) + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                sl[(int64_t)m * (a.NT * 32)] = acc[mr][r];
            }
        }
    }
}

// Sum the S split-K slabs in fixed order (deterministic) and emit f16 (+bias): thread = (row, 4 columns).
__global__ __launch_bounds__(256) void splitk_reduce_f16_kernel(const float* __restrict__ slabs,
                                                                const f16* __restrict__ bias, f16* __restrict__ out,
                                                                int64_t ldo, int M, int N, int NP, int S) {
    const int np4 = NP >> 2;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int mslab = blockIdx.y;
    if (idx >= (int64_t)32 * np4) return;
    const int m = idx / np4, c4 = (idx - (int64_t)m * np4) * 4;
    if (mslab * 32 + m >= M) return;
    f32x4 v = {0, 0, 0, 0};
    const float* base = slabs + ((int64_t)mslab * S * 32 + m) * NP + c4;
    for (int s2 = 0; s2 < S; ++s2) v += *reinterpret_cast<const f32x4*>(base + (int64_t)s2 * 32 * NP);
    f16* o = out + (int64_t)(mslab * 32 + m) * ldo + c4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (c4 + e < N) {
            float f = v[e];
            if (bias) f += (float)bias[c4 + e];
            o[e] = (f16)f;
        }
    }
}

__global__ void gptq_dequant_kernel(const uint8_t* __restrict__ prep, int64_t offB, f16* __restrict__ wout,
                                    int K, int N, int G, int gs, int NT, int KS, int flags) {
    // one thread per prepared int32 (8 k of one column)
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)NT * KS * 256) return;
    int i = idx & 3;
    int l = (idx >> 2) & 63;
    int64_t ks = (idx >> 8) % KS;
    int64_t nt = (idx >> 8) / KS;
    int n = (int)col_src(nt, l & 31, N, flags);
    int k0 = (ks * 8 + (l >> 5) * 4 + i) * 8;
    if (n >= N || k0 >= K) return;
    uint32_t q = reinterpret_cast<const uint32_t*>(prep)[idx];
    int g = min(k0 / gs, G - 1);
    f16x2 szh = __builtin_bit_cast(f16x2, reinterpret_cast<const uint32_t*>(prep + offB)[(nt * G + g) * 32 + (l & 31)]);
    float s = (float)szh[0];
    float z = (float)szh[1] - 1024.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = ((float)((q >> (4 * j)) & 15u) - z) * s;
        wout[(int64_t)(k0 + nib_src(j)) * N + n] = (f16)v;
    }
}

struct GemmPlan {
    int KR, S, WK, TN;  // rows per block, global k splits, in-block k-parts, column tiles per k-part
    int MR;             // 32-row blocks of x per pass (2 for M > 32)
};

// Block = 32*TN columns x KR rows, TN*WK waves (KR a multiple of 256*WK).  Measured on MI355X
// (profiles/r01_gemm_pmc.md): weight streaming alone runs at ~3.3 TB/s for these sizes and dequant + MFMA add on
// top rather than hide, so the plan first spreads the matrix over all 256 CUs (narrow blocks before global
// k-splits, which cost slab traffic), then adds in-block k-parts (free of slab traffic) for waves per SIMD.
static GemmPlan plan_gemm(int64_t K, int64_t N, int act = 0, int64_t M = 32) {
    const int MR = M > 32 ? 2 : 1;
    if (const char* ov = getenv("TGIS_GPTQ_PLAN")) {  // tuning hook: "KR,S,WK,TN"
        int kr = 0, sp = 0, wk = 0, tn = 0;
        if (sscanf(ov, "%d,%d,%d,%d", &kr, &sp, &wk, &tn) == 4 && kr > 0 && (wk == 2 || wk == 4) &&
            (tn >= 2 && tn <= 4) && kr % (KC * wk) == 0 && (int64_t)sp * kr >= K &&
            (int64_t)(sp - 1) * kr < K && (act != 2 || sp == 1) && (MR == 1 || wk == 2))
            return {kr, sp, wk, tn, MR};
    }
    const int64_t tiles = cdiv64(N, 32);
    const int64_t kchunks = cdiv64(K, KC);
    // Rules from the MI355X sweeps in profiles/r01_gemm_pmc.md (tools/sweep_gptq.py), M = 32:
    //  wide N (or the SiLU epilogue, which needs the whole sum in one block): no global split, 96-column blocks of
    //    12 waves while they fit one per CU (gate_up 4096x22016: 230 blocks, 17.3 us vs 19.2 for 128-column blocks);
    //  medium N: 128-column blocks of two k-parts, split K until ~224 blocks (qkv 4096x12288: 10.8 vs 12.6 us);
    //  narrow N: 64-column blocks of four k-parts, split K until 256 blocks.
    //  M > 32 (MR = 2): the x chunk buffers double, so two k-parts per block; 128-column blocks unless N is narrow,
    //    and fewer global splits when several 64-row passes already multiply the blocks.
    int TN, WK;
    int64_t S = 1;
    if (MR == 2) {
        TN = tiles >= 256 ? 4 : 2;
        WK = 2;
        if (act != 2) {
            const int64_t colblocks = cdiv64(tiles, TN) * cdiv64(M, 64);
            S = std::max<int64_t>(1, std::min<int64_t>(kchunks / 2, (224 + colblocks / 2) / colblocks));
            while (S > 1 && (S - 1) * cdiv64(kchunks, S) >= kchunks) --S;
        }
    } else if (act == 2 || tiles >= 512) {
        // narrowest blocks that still fit one per CU (TP shards of gate_up: 344 / 172 / 86 tiles -> 64-column blocks)
        TN = 2;
        while (TN < 4 && cdiv64(tiles, TN) > 256) ++TN;
        WK = 4;
    } else {
        TN = tiles >= 256 ? 4 : 2;
        const int64_t colblocks = cdiv64(tiles, TN);
        const int64_t want = TN == 4 ? 224 : 256;
        S = std::max<int64_t>(1, std::min<int64_t>(kchunks, (want + colblocks / 2) / colblocks));
        while (S > 1 && (S - 1) * cdiv64(kchunks, S) >= kchunks) --S;  // no empty last split
        WK = TN == 4 ? 2 : (cdiv64(kchunks, S) >= 4 ? 4 : 2);
    }
    int64_t KRc = cdiv64(kchunks, S);
    if (KRc < WK) WK = 2;
    KRc = cdiv64(KRc, WK) * WK;  // whole chunks per k-part (rows beyond K contribute zeros)
    while (S > 1 && (S - 1) * KRc >= kchunks) --S;
    return {(int)(KRc * KC), (int)S, WK, TN, MR};
}

// slabs are stored in 32-row units; a 64-row pass always writes both of its units
static int64_t slab_bytes(int64_t M, int64_t N, int S) {
    return S > 1 ? cdiv64(M, 64) * 2 * S * 32 * cdiv64(N, 32) * 32 * 4 : 0;
}

}  // namespace

extern "C" int64_t tgis_gptq_prepared_bytes(int64_t K, int64_t N, int64_t groups) {
    if (K <= 0 || N <= 0 || groups <= 0) return 0;
    return prep_layout(K, N, groups).total;
}

extern "C" int tgis_gptq_prepare(const int32_t* qweight, const int32_t* qzeros, const void* scales,
                                 const int32_t* g_idx_host, int32_t* perm_out, int64_t K, int64_t N,
                                 int64_t groups, int flags, void* prepared, void* stream) {
    TGIS_CHECK_ARG(qweight && qzeros && scales && prepared, "tgis_gptq_prepare: null tensor");
    TGIS_CHECK_ARG(!(flags & 1) || (N % 32 == 0), "tgis_gptq_prepare: gate/up interleave needs N/2 %% 16 == 0");
    TGIS_CHECK_ARG(K > 0 && N > 0 && K % 32 == 0 && N % 32 == 0,
                   "tgis_gptq_prepare: K (%ld) and N (%ld) must be positive multiples of 32", (long)K, (long)N);
    TGIS_CHECK_ARG(groups > 0 && K % groups == 0, "tgis_gptq_prepare: K %% groups != 0");
    const int64_t gs = K / groups;
    TGIS_CHECK_ARG(gs % 8 == 0, "tgis_gptq_prepare: group size %ld not a multiple of 8", (long)gs);
    hipStream_t st = (hipStream_t)stream;
    const int32_t* perm_dev = nullptr;
    if (g_idx_host) {
        bool trivial = true;
        for (int64_t k = 0; k < K; ++k)
            if (g_idx_host[k] != (int32_t)(k / gs)) { trivial = false; break; }
        if (!trivial) {
            TGIS_CHECK_ARG(perm_out, "tgis_gptq_prepare: act-order g_idx needs perm_out");
            std::vector<int32_t> perm(K);
            std::iota(perm.begin(), perm.end(), 0);
            std::stable_sort(perm.begin(), perm.end(),
                             [&](int32_t a, int32_t b) { return g_idx_host[a] < g_idx_host[b]; });
            // every group must own exactly gs rows (true for GPTQ act-order checkpoints)
            for (int64_t k = 0; k < K; ++k)
                TGIS_CHECK_ARG(g_idx_host[perm[k]] == (int32_t)(k / gs),
                               "tgis_gptq_prepare: g_idx groups are not of uniform size");
            TGIS_CHECK_HIP(hipStreamSynchronize(st));
            TGIS_CHECK_HIP(hipMemcpy(perm_out, perm.data(), K * sizeof(int32_t), hipMemcpyHostToDevice));
            perm_dev = perm_out;
        }
    }
    PrepLayout p = prep_layout(K, N, groups);
    uint8_t* base = (uint8_t*)prepared;
    int64_t totalA = p.NT * p.KS * 256;
    hipLaunchKernelGGL(gptq_prepare_w_kernel, dim3((unsigned)cdiv64(totalA, 256)), dim3(256), 0, st, qweight,
                       perm_dev, (int32_t*)base, K, N, p.NT, p.KS, flags);
    TGIS_CHECK_LAUNCH();
    int64_t totalB = p.NT * groups * 32;
    hipLaunchKernelGGL(gptq_prepare_sz_kernel, dim3((unsigned)cdiv64(totalB, 256)), dim3(256), 0, st, qzeros,
                       (const f16*)scales, (uint32_t*)(base + p.offB), N, p.NT, groups, flags);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int64_t tgis_gptq_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N) {
    GemmPlan pl = plan_gemm(K, N, 0, M);
    return 4096 + slab_bytes(M, N, pl.S);
}

template <int TN, int WK, int ACT, bool G64, bool PERM, int MR>
static int launch_one(dim3 grid, size_t lds, hipStream_t st, const GemmArgs& a) {
    static bool attr_done = false;
    if (!attr_done) {
        TGIS_CHECK_HIP(hipFuncSetAttribute((const void*)gptq_gemm_kernel<TN, WK, ACT, G64, PERM, MR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 32 * RS * 2 + 64));
        attr_done = true;
    }
    hipLaunchKernelGGL((gptq_gemm_kernel<TN, WK, ACT, G64, PERM, MR>), grid, dim3(64 * TN * WK), lds, st, a);
    return TGIS_OK;
}
template <int TN, int WK, int ACT, bool G64, bool PERM>
static int launch_variant(int mr, dim3 grid, size_t lds, hipStream_t st, const GemmArgs& a) {
    if constexpr (WK == 2) {  // 64-row passes exist for two-k-part blocks only (LDS)
        if (mr == 2) return launch_one<TN, WK, ACT, G64, PERM, 2>(grid, lds, st, a);
    }
    return launch_one<TN, WK, ACT, G64, PERM, 1>(grid, lds, st, a);
}

static int launch_gptq(const void* x, int64_t ldx, const void* prepared, const void* bias, const int32_t* perm,
                       void* out, int64_t ldo, int64_t M, int64_t K, int64_t N, int64_t groups, int act, float* slabs,
                       int partial, const GemmPlan& pl, hipStream_t st) {
    PrepLayout p = prep_layout(K, N, groups);
    const int64_t mslabs = cdiv64(M, 32 * pl.MR);  // passes over the weights
    const int64_t gs = K / groups;
    const int64_t spg = gs / 64;  // k64-steps per group
    const bool group64 = groups == 1 || (gs % 64 == 0 && (spg & (spg - 1)) == 0);
    GemmArgs a;
    a.x = (const f16*)x;
    a.ldx = ldx;
    a.prep = (const uint8_t*)prepared;
    a.offB = p.offB;
    a.bias = (const f16*)bias;
    a.perm = perm;
    a.out = (f16*)out;
    a.ldo = ldo;
    a.M = (int)M;
    a.K = (int)K;
    a.N = (int)N;
    a.G = (int)groups;
    a.gs = (int)gs;
    a.KR = pl.KR;
    a.S = pl.S;
    a.NT = (int)p.NT;
    a.KS = (int)p.KS;
    a.slabs = slabs;
    a.partial = partial;
    a.spg_shift = 30;
    if (groups > 1 && group64)
        for (a.spg_shift = 0; (1 << a.spg_shift) < spg; ++a.spg_shift) {}
    dim3 grid((unsigned)cdiv64(p.NT, pl.TN), (unsigned)pl.S, (unsigned)mslabs);
    const size_t lds = (size_t)pl.WK * 2 * 32 * pl.MR * RS * sizeof(f16) + 64;  // x buffers + arrival counters
#define TGIS_LAUNCH_GEMM(T, W, A, G, P)                                                        \
    do {                                                                                       \
        int rc_ = launch_variant<T, W, A, G, P>(pl.MR, grid, lds, st, a);                      \
        if (rc_ != TGIS_OK) return rc_;                                                        \
    } while (0)
#define TGIS_LAUNCH_GEMM_W(A, G, P)                      \
    do {                                                 \
        const int tw = pl.TN * 10 + pl.WK;               \
        if (tw == 44)                                    \
            TGIS_LAUNCH_GEMM(4, 4, A, G, P);             \
        else if (tw == 42)                               \
            TGIS_LAUNCH_GEMM(4, 2, A, G, P);             \
        else if (tw == 34)                               \
            TGIS_LAUNCH_GEMM(3, 4, A, G, P);             \
        else if (tw == 32)                               \
            TGIS_LAUNCH_GEMM(3, 2, A, G, P);             \
        else if (tw == 24)                               \
            TGIS_LAUNCH_GEMM(2, 4, A, G, P);             \
        else                                             \
            TGIS_LAUNCH_GEMM(2, 2, A, G, P);             \
    } while (0)
    const int variant = (act == 1 ? 4 : act == 2 ? 8 : 0) | (group64 ? 2 : 0) | (perm ? 1 : 0);
    switch (variant) {
        case 8: TGIS_LAUNCH_GEMM_W(2, false, false); break;
        case 9: TGIS_LAUNCH_GEMM_W(2, false, true); break;
        case 10: TGIS_LAUNCH_GEMM_W(2, true, false); break;
        case 11: TGIS_LAUNCH_GEMM_W(2, true, true); break;
        case 0: TGIS_LAUNCH_GEMM_W(0, false, false); break;
        case 1: TGIS_LAUNCH_GEMM_W(0, false, true); break;
        case 2: TGIS_LAUNCH_GEMM_W(0, true, false); break;
        case 3: TGIS_LAUNCH_GEMM_W(0, true, true); break;
        case 4: TGIS_LAUNCH_GEMM_W(1, false, false); break;
        case 5: TGIS_LAUNCH_GEMM_W(1, false, true); break;
        case 6: TGIS_LAUNCH_GEMM_W(1, true, false); break;
        case 7: TGIS_LAUNCH_GEMM_W(1, true, true); break;
    }
#undef TGIS_LAUNCH_GEMM_W
#undef TGIS_LAUNCH_GEMM
    TGIS_CHECK_LAUNCH();
    if (!partial && pl.S > 1 && !getenv("TGIS_GPTQ_NOREDUCE")) {
        const int NP = (int)p.NT * 32;
        dim3 rgrid((unsigned)cdiv64((int64_t)32 * (NP / 4), 256), (unsigned)cdiv64(M, 32));
        hipLaunchKernelGGL(splitk_reduce_f16_kernel, rgrid, dim3(256), 0, st, a.slabs, a.bias, a.out, a.ldo, a.M, a.N,
                           NP, a.S);
        TGIS_CHECK_LAUNCH();
    }
    return TGIS_OK;
}

static int check_gemm_args(const void* x, int64_t ldx, const void* prepared, int64_t M, int64_t K, int64_t N,
                           int64_t groups, int act) {
    TGIS_CHECK_ARG(x && prepared, "tgis_gptq_gemm: null tensor");
    TGIS_CHECK_ARG(M >= 0 && K > 0 && N > 0 && K % 32 == 0 && N % 32 == 0, "tgis_gptq_gemm: bad shape");
    TGIS_CHECK_ARG(groups > 0 && K % groups == 0, "tgis_gptq_gemm: K %% groups != 0");
    TGIS_CHECK_ARG(act == 0 || act == 1 || act == 2, "tgis_gptq_gemm: act must be 0, 1 or 2");
    TGIS_CHECK_ARG(act != 2 || N % 32 == 0, "tgis_gptq_gemm: act=2 needs N/2 to be a multiple of 16");
    TGIS_CHECK_ARG(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0, "tgis_gptq_gemm: x must be 16-byte aligned rows");
    return TGIS_OK;
}

extern "C" int tgis_gptq_gemm_f16(const void* x, int64_t ldx, const void* prepared, const void* bias,
                                  const int32_t* perm, void* out, int64_t ldo, int64_t M, int64_t K,
                                  int64_t N, int64_t groups, int act, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    int rc = check_gemm_args(x, ldx, prepared, M, K, N, groups, act);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(out, "tgis_gptq_gemm_f16: null out");
    if (M == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    GemmPlan pl = plan_gemm(K, N, act, M);
    TGIS_CHECK_ARG(cdiv64(M, 32) <= 65535, "tgis_gptq_gemm_f16: M too large for one launch");
    int64_t need = 4096 + slab_bytes(M, N, pl.S);
    TGIS_CHECK_ARG(workspace && workspace_bytes >= need, "tgis_gptq_gemm_f16: workspace too small (%ld < %ld)",
                   (long)workspace_bytes, (long)need);
    TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
    return launch_gptq(x, ldx, prepared, bias, perm, out, ldo, M, K, N, groups, act,
                       (float*)((uint8_t*)workspace + 4096), 0, pl, st);
}

extern "C" int64_t tgis_gptq_gemm_partial_bytes(int64_t M, int64_t K, int64_t N) {
    GemmPlan pl = plan_gemm(K, N, 0, M);
    return cdiv64(std::max<int64_t>(M, 1), 64) * 2 * pl.S * 32 * cdiv64(N, 32) * 32 * 4;
}

extern "C" int tgis_gptq_gemm_f16_partial(const void* x, int64_t ldx, const void* prepared, const int32_t* perm,
                                          int64_t M, int64_t K, int64_t N, int64_t groups, int act, float* slabs,
                                          int64_t slabs_bytes, int* num_slabs, int64_t* slab_ld, void* stream) {
    int rc = check_gemm_args(x, ldx, prepared, M, K, N, groups, act);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(M >= 1 && cdiv64(M, 32) <= 65535, "tgis_gptq_gemm_f16_partial: bad M");
    TGIS_CHECK_ARG(act != 2, "tgis_gptq_gemm_f16_partial: act=2 cannot be deferred");
    GemmPlan pl = plan_gemm(K, N, 0, M);
    TGIS_CHECK_ARG(slabs && slabs_bytes >= tgis_gptq_gemm_partial_bytes(M, K, N),
                   "tgis_gptq_gemm_f16_partial: slab buffer too small");
    hipStream_t st = (hipStream_t)stream;
    if (num_slabs) *num_slabs = pl.S;
    if (slab_ld) *slab_ld = cdiv64(N, 32) * 32;
    TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
    return launch_gptq(x, ldx, prepared, nullptr, perm, nullptr, 0, M, K, N, groups, act, slabs, 1, pl, st);
}

// debug aid (not part of the documented ABI): resident blocks per CU the runtime reports for the main kernel
#ifdef TGIS_TRACE
extern "C" int tgis_debug_set_trace(void* ptr) {
    long long* p = (long long*)ptr;
    TGIS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)));
    return TGIS_OK;
}
#endif

extern "C" int tgis_debug_gemm_occupancy(int tn, int wk) {
    int nb = -1;
    const size_t lds = (size_t)wk * 2 * 32 * RS * sizeof(f16) + 64;
    if (tn == 4 && wk == 4)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gptq_gemm_kernel<4, 4, 0, true, false, 1>, 1024, lds);
    else if (tn == 2 && wk == 4)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gptq_gemm_kernel<2, 4, 0, true, false, 1>, 512, lds);
    else if (tn == 2 && wk == 2)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gptq_gemm_kernel<2, 2, 0, true, false, 1>, 256, lds);
    else if (tn == 3 && wk == 4)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gptq_gemm_kernel<3, 4, 0, true, false, 1>, 768, lds);
    return nb;
}

extern "C" int tgis_gptq_dequant_f16(const void* prepared, void* w_out, int64_t K, int64_t N, int64_t groups,
                                     int flags, void* stream) {
    TGIS_CHECK_ARG(prepared && w_out && K > 0 && N > 0 && groups > 0 && K % groups == 0,
                   "tgis_gptq_dequant_f16: bad arguments");
    PrepLayout p = prep_layout(K, N, groups);
    int64_t total = p.NT * p.KS * 256;
    hipLaunchKernelGGL(gptq_dequant_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)prepared, p.offB, (f16*)w_out, (int)K, (int)N, (int)groups,
                       (int)(K / groups), (int)p.NT, (int)p.KS, flags);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}
