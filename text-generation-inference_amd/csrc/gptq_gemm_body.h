// The int4 GPTQ streaming GEMM as a device function (one workgroup = one unit) plus its launch planning; the kernel that
// runs it is in gptq.hip.  (Rounds 2 - 5 also ran this unit inside a persistent "decode tail" launch and with an add +
// RMSNorm phase in front of it — both measured slower than separate launches and removed in round 6: experiments/README.md.)
#pragma once
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "kv_layout.h"

namespace gptq {

struct PrepLayout {
    int64_t NT, KS, G, offB, total;
};
static inline PrepLayout prep_layout(int64_t K, int64_t N, int64_t G) {
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
    // ACT == 3 (rope image): the epilogue rotates q / k heads and writes k / v into their cache pages; `out` is the q tensor
    const int32_t* positions;  // [M]
    const int32_t* slots;      // [M] page * 32 + token
    const f16* cosb;           // [max_pos][rD / 2]
    const f16* sinb;
    f16* kpool;                // [pages][rHkv][32 * rD] in the K page layout of kv_layout.h
    f16* vpool;
    int rH, rHkv, rD;
};

#ifdef TGIS_TRACE
static __device__ long long* g_trace = nullptr;  // [blocks][16 waves][32 stamps] of s_memtime (debug builds only)
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
// LDS of one unit: the x chunk buffers of the WK k-parts (reused by the k-part reduction), then the k-part arrival counters.

typedef __attribute__((address_space(3))) int lds_int;  // explicit LDS pointer: a generic one costs vmcnt(0) waits

// A wave keeps RING = 4 one-KiB weight loads (one chunk) and the {scale, zero} words of their k64-steps in flight; a
// two-chunk ring measured ~1 us SLOWER on every cfg3 shape (the first barrier waits for twice the prologue loads to issue).
template <int TN, int WK, int ACT, bool GROUP64, bool PERM, int MR>
__device__ __forceinline__ void gptq_gemm_unit(const GemmArgs& a, const int ntg, const int split, const int mslab,
                                               unsigned char* smem) {
    static_assert(MR == 1 || WK == 2, "64-row passes need the LDS of two k-parts");
    static_assert(WK > 1, "the finish below exchanges k-parts");
    static_assert(ACT != 3 || !PERM, "the rope epilogue is a decode form (<= 64 rows, no act-order)");
    constexpr int RING = 4;
    constexpr int XR = 32 * MR;                 // x rows per pass
    constexpr int GT = 64 * TN;                 // threads of one k-part group
    constexpr int NJ = (XR * 32 + GT - 1) / GT;  // 16-byte x pieces per thread per chunk (XR rows x 32 pieces per chunk)
    constexpr int RSTEP = GT / 32;              // rows covered by one pass of the group
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index and everything derived from it (tile, k-range, image addresses) is wave-uniform: keep it in
    // SGPRs so that the address/clamp arithmetic of the loop runs on the scalar unit, not on the VALU that the
    // dequantisation saturates
    TRACE(0);
    TRACE_RT(14);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // k-part = w % WK, so that every k-part owns waves of every age.  A CU serves its older waves first
    // (issue and memory return): with k-part = w / TN the k-part of the youngest waves finished 2.5 us (gate_up) after the
    // k-part of the oldest ones, and everybody waited for it; the per-chunk sync of a mixed k-part holds its old waves back
    // instead, which is what lets the young ones catch up.
    const int wn = w / WK, wk = w % WK, ltid = wn * 64 + lane;
    f16* xs = reinterpret_cast<f16*>(smem) + wk * (2 * XR * RS);  // this k-part's [2][XR][RS]
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
    u32x4 wq[RING];
    uint32_t szr[RING];

    // ACT 3: cache slot and rotary position of the rows this wave will finish (see the distributed epilogue)
    int32_t rpos[ACT == 3 ? MR : 1][ACT == 3 ? 16 / WK : 1], rslot[ACT == 3 ? MR : 1][ACT == 3 ? 16 / WK : 1];
    if (ACT == 3) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int j = 0; j < 16 / WK; ++j) {
                const int r = wk * (16 / WK) + j;
                const int m = min(mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), mrows - 1);
                rpos[mr][j] = a.positions[m0 + m];
                rslot[mr][j] = a.slots[m0 + m];
            }
    }

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
                    const int ksrc = a.perm[kc + e];  // -1: a pad row of an act-order row shard (utils/weights.py) reads 0
                    v[e] = ksrc >= 0 ? xr[ksrc] : (f16)0.f;
                    if (ACT == 1) u[e] = ksrc >= 0 ? xr[a.K + ksrc] : (f16)0.f;
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
    // One accumulator: the ~15 dequantisation instructions between two MFMAs already cover the 32-cycle dependency of
    // consecutive MFMAs on it, and the 16 registers a second (alternating) accumulator took are worth more as occupancy
    // (round 2, cfg3 decode: 4.93 -> 4.84 ms/step).
    constexpr int NACC = 1;
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
    volatile lds_int* sync_cnt = (volatile lds_int*)(smem + (size_t)WK * 2 * XR * RS * sizeof(f16)) + wk;
    // all waves of the unit meet here; the LDS traffic a wave issued before is complete when it arrives
    auto unit_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (wn == 0 && lane == 0) *sync_cnt = 0;
    // Issue order matters: a wave's loads return in order, so the (L2-resident) first x chunk goes out before the
    // HBM weight stream it would otherwise queue behind; then the small scale loads, then the ring of weights.
    stage_load(0);
    // The block-wide barrier that publishes the zeroed counters sits HERE, between the x requests and the weight requests
    // of every wave.  A CU serves its waves' requests in arrival order: without it a later wave's first x chunk queues
    // behind the HBM weight requests of the waves that started before it, and the first chunk was staged only when the
    // whole first ring had arrived (3.1 us after entry for gate_up); with it every x request of the block is in front of
    // every weight request.  From here on each k-part group paces itself.
    unit_barrier();
#pragma unroll
    for (int s = 0; s < RING; ++s)
        if (GROUP64) szr[s] = sz_at(s);
#pragma unroll
    for (int s = 0; s < RING; ++s) wq[s] = w_at(s);
    auto group_sync = [&](int target) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lds_int*)sync_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__builtin_amdgcn_readfirstlane(*sync_cnt) < target) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    stage_store(0);
    group_sync(TN);
    TRACE(2);

    // One chunk = 4 k64-steps.  All but the last chunk prefetch: next chunk's x into registers, next chunk's
    // scales, and each weight slot is refilled in place right after it is consumed.  The last chunk is a separate
    // instantiation without any of that (no wasted re-reads, no sync).
    // SB: first ring slot of this chunk; STAGE: a next chunk exists (prefetch its x, sync at the end); REFILL: the chunk
    // RING / 4 ahead exists (its weights replace this chunk's in place).  All three are compile-time: the loop stays
    // branch-free around the loads.
    auto chunk_body = [&](const int chunk, auto sb_tag, auto stage_tag, auto refill_tag) {
        constexpr int SB = decltype(sb_tag)::value;
        constexpr bool STAGE = decltype(stage_tag)::value, REFILL = decltype(refill_tag)::value;
        if (STAGE) stage_load(chunk + 1);
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
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = dequant8(cur[i], zc, zd, sc, EXr, M0r, M1r);
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
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    f16x8 av = ld16<f16x8>(xk + mr * (32 * RS) + i * 8);
                    accs[mr][i & (NACC - 1)] = mfma32(av, b[i], accs[mr][i & (NACC - 1)]);
                }
            }
        }
        if (GROUP64 && REFILL) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) szr[SB + s4] = szn[s4];
        }
        if (!STAGE) return;
        stage_store((chunk + 1) & 1);
        TRACE(3 + 2 * min(chunk, 3));
        group_sync(TN * (chunk + 2));
        TRACE(4 + 2 * min(chunk, 3));
    };
    using I0 = std::integral_constant<int, 0>;
    using Y = std::true_type;
    using N = std::false_type;
    for (int chunk = 0; chunk + 1 < nchunks; ++chunk) chunk_body(chunk, I0{}, Y{}, Y{});
    chunk_body(nchunks - 1, I0{}, N{}, N{});
    TRACE(9);
    // The unit finishes DISTRIBUTED (below): wave (wn, wk) ends up with accumulator registers [wk NR, (wk + 1) NR) of
    // its tile, i.e. NR of the 16 row groups.  ACT 3: it asks for those rows' cos / sin entries now (the positions were
    // loaded at entry), so that the round trip runs under the k-part exchange.
    constexpr int NR = 16 / WK;
    f16 rcos[ACT == 3 ? MR : 1][ACT == 3 ? NR : 1], rsin[ACT == 3 ? MR : 1][ACT == 3 ? NR : 1];
    if (ACT == 3) {
        const int per = a.rD >> 5;
        const int tt = nt - (nt / per) * per;
        const int dr = 16 * tt + (lane & 15);
        const bool roth = nt / per < a.rH + a.rHkv;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                rcos[mr][j] = roth ? a.cosb[(int64_t)rpos[mr][j] * (a.rD >> 1) + dr] : (f16)1.f;
                rsin[mr][j] = roth ? a.sinb[(int64_t)rpos[mr][j] * (a.rD >> 1) + dr] : (f16)0.f;
            }
    }
    unit_barrier();  // every k-part is done with its x buffers: the reduction below reuses them

    f32x16 acc[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) acc[mr] = NACC == 2 ? accs[mr][0] + accs[mr][1] : accs[mr][0];
    // ---- distributed finish ------------------------------------------------------------------------------------------
    // Every wave (k-part 0 included) leaves its partial sums in LDS; wave (wn, wk) then sums the WK k-parts of registers
    // [wk NR, (wk + 1) NR) of tile wn in the fixed order 0..WK-1 (bit-identical to the reducer-wave form) and runs the
    // epilogue for those rows only: the loads, the arithmetic and the scattered stores of the tail are spread over all
    // waves of the block instead of a quarter of them.
    {
        float* red = reinterpret_cast<float*>(smem);  // [WK][TN tiles][MR][64 lanes][16]; the x buffers are dead now
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            float* dst = red + ((((wk * TN + wn) * MR + mr) * 64 + lane) << 4);
#pragma unroll
            for (int r = 0; r < 16; r += 4)
                *reinterpret_cast<f32x4*>(dst + r) = f32x4{acc[mr][r], acc[mr][r + 1], acc[mr][r + 2], acc[mr][r + 3]};
        }
        unit_barrier();
        TRACE(11);
        if (nt_raw >= a.NT) return;
        float fin[MR][NR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int j = 0; j < NR; ++j) fin[mr][j] = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < WK; ++k2) {
                const float* src = red + ((((k2 * TN + wn) * MR + mr) * 64 + lane) << 4) + wk * NR;
#pragma unroll
                for (int j = 0; j < NR; j += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(src + j);
                    if (k2 == 0) {
                        fin[mr][j] = t[0], fin[mr][j + 1] = t[1], fin[mr][j + 2] = t[2], fin[mr][j + 3] = t[3];
                    } else {
                        fin[mr][j] += t[0], fin[mr][j + 1] += t[1], fin[mr][j + 2] += t[2], fin[mr][j + 3] += t[3];
                    }
                }
            }
        }
        TRACE(12);
        const int c = lane & 31;
        auto row_of = [&](int j) {  // row (within the 32-row block) of finished register j
            const int r = wk * NR + j;
            return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        };
        if (ACT == 3) {
            // Rope image (col_src flags bit 1): a tile of a q or k head holds dims [16 t, 16 t + 16) in lanes c < 16 and
            // their rotation partners rD/2 + [16 t, 16 t + 16) in lanes c + 16; v heads keep 32 consecutive dims.  The sum
            // (+ bias) is rounded to f16 first, then rotated in fp32 — the arithmetic of rope_kv_kernel on the reduced
            // activation (rotary_emb.apply_rotary, utils/layers.py:466-472) — and q goes to `out`, k / v straight into
            // their cache pages (flash_llama_modeling.py:268,282).  Host guarantees S == 1.
            const int per = a.rD >> 5;
            const int head = nt / per, tt = nt - head * per;
            const bool roth = head < a.rH + a.rHkv;
            const int d = roth ? ((c < 16) ? 16 * tt + c : (a.rD >> 1) + 16 * tt + (c - 16)) : 32 * tt + c;
            const int col = head * a.rD + d;
            const float bv = a.bias ? (float)a.bias[col] : 0.f;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int m = mr * 32 + row_of(j);
                const float mine = (float)(f16)(fin[mr][j] + bv);
                float o = mine;
                if (roth) {
                    const float other = __shfl_xor(mine, 16, 64);
                    const float cf = (float)rcos[mr][j], sf = (float)rsin[mr][j];
                    o = (c < 16) ? mine * cf - other * sf : other * sf + mine * cf;
                }
                const f16 oh = (f16)o;
                if (m < mrows) {
                    if (head < a.rH) {
                        a.out[(int64_t)(m0 + m) * a.ldo + col] = oh;
                    } else {
                        const int page = rslot[mr][j] >> 5, tok = rslot[mr][j] & 31;
                        if (roth)
                            a.kpool[((int64_t)page * a.rHkv + (head - a.rH)) * 32 * a.rD + k_off(tok, d, a.rD)] = oh;
                        else
                            a.vpool[((int64_t)page * a.rHkv + (head - a.rH - a.rHkv)) * 32 * a.rD + v_off(tok, d, a.rD)] = oh;
                    }
                }
            }
            return;
        }
        const int n = nt * 32 + c;
        if (ACT == 2) {
            // interleaved gate/up pairs (col_src flags bit 0): lanes c < 16 hold gate column j2 = 16 nt + c, lanes c + 16
            // the matching up column.  out[m][j2] = f16(f16(silu(f16 gate)) * f16 up), the rounding sequence of the
            // reference's eager ops (flash_llama_modeling.py:332-335).  Host guarantees S == 1.
            const int half = a.N >> 1;
            const int j2 = nt * 16 + (c & 15);
            const int nsrc = (c < 16) ? j2 : half + j2;
            const float bv = (a.bias && j2 < half) ? (float)a.bias[nsrc] : 0.f;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const float mine = (float)(f16)(fin[mr][j] + bv);
                    const float other = __shfl_xor(mine, 16, 64);
                    const int m = mr * 32 + row_of(j);
                    if (c < 16 && j2 < half && m < mrows) {
                        float sl = mine / (1.f + __expf(-mine));
                        a.out[(int64_t)(m0 + m) * a.ldo + j2] = (f16)((float)(f16)sl * other);
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
                    for (int j = 0; j < NR; ++j) {
                        const int m = mr * 32 + row_of(j);
                        if (m < mrows) a.out[(int64_t)(m0 + m) * a.ldo + n] = (f16)(fin[mr][j] + bv);
                    }
            }
        } else {
            // slabs are indexed in 32-row units: this pass owns units mslab*MR .. mslab*MR + MR-1
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                float* sl = a.slabs + ((int64_t)((mslab * MR + mr) * a.S + split) * 32) * (a.NT * 32) + n;
#pragma unroll
                for (int j = 0; j < NR; ++j) sl[(int64_t)row_of(j) * (a.NT * 32)] = fin[mr][j];
            }
        }
        return;
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
static inline GemmPlan plan_gemm(int64_t K, int64_t N, int act = 0, int64_t M = 32) {
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

// sum of the S split-K slabs (+ bias) -> f16, fixed order (gptq.hip)
int reduce_slabs(const float* slabs, const f16* bias, f16* out, int64_t ldo, int M, int N, int NP, int S, hipStream_t st);

// slabs are stored in 32-row units; a 64-row pass always writes both of its units
static inline int64_t slab_bytes(int64_t M, int64_t N, int S) {
    return S > 1 ? cdiv64(M, 64) * 2 * S * 32 * cdiv64(N, 32) * 32 * 4 : 0;
}


}  // namespace gptq
