// Loader / consumer form of the lean int4 decode GEMM (round 3).
//
// gptq_gemm_unit / gptq_lean_unit keep the weight stream in registers: each wave has 4 KiB in flight and re-requests a
// slot only after it has consumed it, so a CU holds at most 48-64 KB of dependent requests and the loop runs at ~3.5 TB/s
// whatever the arithmetic costs (measured: the lean arithmetic, 60 % fewer VALU operations, changed nothing).  Here ONE
// loader wave per workgroup streams the 1 KiB wave-steps of ALL consumer waves into an LDS ring with
// `global_load_lds_dwordx4 ... nt` (LDS-DMA: no VGPRs, no dependence on the consumers except ring space), throttled by a
// counted vmcnt; tools/floor/ldsdma.hip: one such wave per CU sustains 5.6-6.0 TB/s chip-wide from 16 KiB in flight.
// The consumer waves (TN column tiles x WK k-parts, as before) read their steps from the ring (one conflict-free
// ds_read_b128 per step), stage x and the row-sum fragments exactly like gptq_lean_unit, and never touch the weight
// image themselves.
//
//   landed   (LDS word, written by the loader): number of DMAs, in issue order, known to have landed.  Issue order is
//            round-robin: round r = step r of every consumer, sequence number r * NC + c.
//   consumed (LDS word per consumer): steps that consumer has copied out of the ring.
// Every spin is bounded; a give-up leaves a code in `*err` (results are then garbage, the launch still ends).
#pragma once
#include "gptq_lean_body.h"

namespace gptq {

// Measurement switches of tools/lean_gemm.py builds (-DLEAN_DBG=bits, wrong results): 1 no x loads, 2 no row-sum loads,
// 4 no scale loads, 8 consumers do not wait for / read fresh ring data, 16 the loader issues nothing.
#ifndef LEAN_DBG
#define LEAN_DBG 0
#endif
constexpr int LD_DBG = LEAN_DBG;

constexpr int KCL = 128;      // k per x chunk = one group = two wave-steps
constexpr int RSL = KCL + 8;  // LDS row stride in halves (+16 B: conflict-free ds_read_b128)
constexpr int LD_XBYTES = 2 * 32 * RSL * (int)sizeof(f16);  // x chunk double buffer of one k-part
constexpr int LD_APBYTES = 2 * 32 * 16;                     // A' fragments [buf][row][8 halves]
constexpr unsigned LD_SPIN_LIMIT = 1u << 22;
__host__ __device__ constexpr int ld_lds_bytes(int NC, int WK, int D) {
    return NC * D * 1024 + WK * (LD_XBYTES + LD_APBYTES) + 256;
}

template <int NT>
__device__ __forceinline__ void dma1k(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    // M0 carries the LDS destination and is compiler-reserved: saved and restored inside the statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int TN, int WK, int ACT, int D>
__device__ __forceinline__ void gptq_ld_unit(const LeanArgs& la, const int ntg, const int split, unsigned char* smem) {
    static_assert(D == 4 || D == 8, "ring slots per consumer");
    static_assert(ACT == 0 || ACT == 2, "plain or SiLU(gate) * up epilogue");
    const GemmArgs& a = la.g;
    constexpr int NC = TN * WK;                      // consumer waves; wave 0 of the workgroup is the loader
    static_assert(NC <= 15, "1024 threads per workgroup");
    // DMAs the loader keeps in flight.  Its progress report lags by VMAX, and a consumer takes two rounds at a time, so the
    // report reaches a waiting consumer without a drain only if VMAX <= (D - 2) NC.
    constexpr int VMAX = (D - 2) * NC < 32 ? (D - 2) * NC : 32;
    constexpr int GT = 64 * TN;
    constexpr int NJ = (32 * 16 + GT - 1) / GT;      // 16-byte x pieces per thread per chunk (32 rows x 16 pieces)
    constexpr int RING_BYTES = NC * D * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* xreg = smem + RING_BYTES;
    unsigned char* apreg = xreg + WK * LD_XBYTES;
    unsigned char* ctrl = apreg + WK * LD_APBYTES;   // [0,16): zeros; +16: WK arrival counters; +32: landed; +64: consumed[NC]
    volatile lds_int* landed = (volatile lds_int*)(ctrl + 32);
    volatile lds_int* consumed = (volatile lds_int*)(ctrl + 64);
    const int mrows = min(32, a.M);
    const int krp = a.KR / WK;                       // rows per k-part (multiple of 256)
    const int nsteps = krp >> 6;
    const int nchunks = krp / KCL;
    const int ks_last = a.KS - 2;

    TRACE_RT(0);
    if (tid < 64) reinterpret_cast<uint32_t*>(ctrl)[tid] = 0u;  // zero line, counters, landed, consumed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    TRACE_RT(1);

    if (w == 0) {
        // ================================ loader ================================
        const unsigned ring = (unsigned)(size_t)smem;  // LDS byte address of the ring
        const char* tbase[NC];
        int kfirst[WK], kclamp[WK];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int nt = min(ntg * TN + (c % TN), a.NT - 1);
            tbase[c] = reinterpret_cast<const char*>(a.prep) + (int64_t)nt * a.KS * 1024;
        }
#pragma unroll
        for (int k = 0; k < WK; ++k) {
            const int k0 = split * a.KR + k * krp, k1 = min(a.K, k0 + krp);
            kfirst[k] = k0 >> 6;
            kclamp[k] = min(ks_last, max(k0 >> 6, ((k1 + 63) >> 6) - 1));
        }
        const uint32_t woff = lane * 16;
        if (LD_DBG & 16) {
            if (lane == 0) *landed = nsteps * NC;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (WK > 1) __builtin_amdgcn_s_barrier();
            return;
        }
        int minc = 0, pub = 0;  // slowest consumer as last seen; progress last reported (monotonic)
        for (int r = 0; r < nsteps; ++r) {
            if (r >= D && minc + D <= r) {
                auto slowest = [&]() -> int {  // steps the slowest consumer has copied out of the ring
                    int v = lane < NC ? consumed[lane] : 0x7fffffff;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
                    return __builtin_amdgcn_readfirstlane(v);
                };
                minc = slowest();
                if (minc + D <= r) {
                    // the ring really is full: everything issued so far is wanted NOW — let it land, say so, and wait for
                    // the slowest consumer to free round r's slots
                    wait_vm<0>();
                    pub = r * NC;
                    if (lane == 0) *landed = pub;
                    for (unsigned spins = 0;; ++spins) {
                        minc = slowest();
                        if (minc + D > r) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (spins > LD_SPIN_LIMIT) {
                            if (a.err && lane == 0)
                                __hip_atomic_store(a.err, 11u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
            }
            const unsigned slot = ring + (unsigned)(r & (D - 1)) * 1024;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ks = min(kfirst[c / TN] + r, kclamp[c / TN]);
                const char* p = tbase[c] + (int64_t)ks * 1024;
                dma1k<1>(p + woff, __builtin_amdgcn_readfirstlane(slot + (unsigned)c * (D * 1024)));
                wait_vm<VMAX>();
            }
            if (r == 0) TRACE_RT(2);
            const int known = (r + 1) * NC - VMAX;
            if (known > pub) {
                pub = known;
                if (lane == 0) *landed = known;
            }
        }
        // tail: publish the last VMAX in a few steps
        TRACE_RT(3);
        const int total = nsteps * NC;
        if (VMAX > 16) { wait_vm<16>(); if (lane == 0) *landed = total - 16; }
        if (VMAX > 8) { wait_vm<8>(); if (lane == 0) *landed = total - 8; }
        wait_vm<0>();
        if (lane == 0) *landed = total;
        TRACE_RT(4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // (2) consumers are done with ring and x buffers
        TRACE_RT(6);
        if (WK > 1) __builtin_amdgcn_s_barrier();  // (3) k-part partials are in LDS
        return;
    }

    // ================================ consumers ================================
    const int c = w - 1;
    const int wn = c % TN, wk = c / TN, ltid = wn * 64 + lane;
    f16* xs = reinterpret_cast<f16*>(xreg + wk * LD_XBYTES);   // [2][32][RSL]
    unsigned char* aps = apreg + wk * LD_APBYTES;              // [2][32][16 B]
    const int k0 = split * a.KR + wk * krp;
    const int k1 = min(a.K, k0 + krp);
    const int nt_raw = ntg * TN + wn;
    const int nt = min(nt_raw, a.NT - 1);
    const char* sztile = reinterpret_cast<const char*>(a.prep + a.offB) + (int64_t)nt * a.G * 128;
    const uint32_t szoff = (lane & 31) * 4;
    const int g0 = k0 >> 7, g_end = k1 >> 7;
    auto sz_at = [&](int grp) -> uint32_t {
        const char* p = sztile + (int64_t)min(g0 + grp, a.G - 1) * 128;
        PIN_SGPR(p);
        if (LD_DBG & 4) return 0x64003c00u;
        const uint32_t v = *(const GLOBAL_AS uint32_t*)(p + szoff);
        return g0 + grp < g_end ? v : 0u;
    };

    // x staging: piece p = ltid + j GT: row p >> 4, 16-byte column piece p & 15; row sums: quarter-task q = ltid < 128:
    // row q >> 2, quarter q & 3 (two of the chunk's eight 16-k blocks)
    f16x8 xg[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) xg[j] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t rowoff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rowoff[j] = (uint32_t)(min((ltid + j * GT) >> 4, mrows - 1) * (int)a.ldx * 2);
    const int pcol = (ltid & 15) * 8;  // GT is a multiple of 16: the column piece does not depend on j
    f32x4 xq = {0, 0, 0, 0};
    const uint32_t xsrow = (uint32_t)(min(ltid >> 2, mrows - 1) * (int)la.ldxs);
    const int kb_last = (a.K >> 4) - 2;
    auto stage_load = [&](int chunk) {
        const int kc = min(k0 + chunk * KCL + pcol, a.K - 8);
        const char* xb = reinterpret_cast<const char*>(a.x);
        PIN_SGPR(xb);
        if (!(LD_DBG & 1)) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) xg[j] = *(const GLOBAL_AS f16x8*)(xb + rowoff[j] + (uint32_t)kc * 2);
        }
        const char* sb = reinterpret_cast<const char*>(la.xs);
        PIN_SGPR(sb);
        const int kb = min(((k0 + chunk * KCL) >> 4) + (ltid & 3) * 2, kb_last);
        if (!(LD_DBG & 2)) xq = *(const GLOBAL_AS f32x4*)(sb + ((size_t)xsrow + (uint32_t)kb) * 8);
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int p = ltid + j * GT;
            if (NJ * GT == 512 || p < 512) st16(xs + buf * (32 * RSL) + (p >> 4) * RSL + pcol, xg[j]);
        }
        float xa = xq[0] + xq[2], xb = xq[1] + xq[3];
        xa += dpp_quad_swap1(xa);
        xb += dpp_quad_swap1(xb);
        xa += dpp_quad_swap2(xa);
        xb += dpp_quad_swap2(xb);
        f16 ah, am, al, bh, bm, bl;
        split3(xa * 0.0625f, ah, am, al);
        split3(xb * 0.0625f, bh, bm, bl);
        const f16x8 ap = {ah, am, al, bh, bm, bl, (f16)0.f, (f16)0.f};
        if ((ltid & 3) == 0 && ltid < 128) st16(aps + (buf * 32 + (ltid >> 2)) * 16, ap);
    };

    uint32_t EXA = 0x64006400u, EXB = 0x54005400u, M0r = 0x000F000Fu, M1r = 0x00F000F0u;
    asm volatile("" : "+v"(EXA), "+v"(EXB));
    asm volatile("" : "+s"(M0r), "+s"(M1r));
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc = zero16;
    const int xoff = (lane & 31) * RSL + (lane >> 5) * 32;
    const unsigned char* apmine = lane < 32 ? aps + lane * 16 : ctrl;   // lanes 32..63 (k slots 8..15) read zeros
    const int apstride = lane < 32 ? 32 * 16 : 0;
    const unsigned char* myring = smem + (size_t)c * (D * 1024) + lane * 16;
    volatile lds_int* sync_cnt = (volatile lds_int*)(ctrl + 16) + wk;
    auto group_sync = [&](int target) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lds_int*)sync_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (unsigned spins = 0; __builtin_amdgcn_readfirstlane(*sync_cnt) < target; ++spins) {
            __builtin_amdgcn_s_sleep(1);
            if (spins > LD_SPIN_LIMIT) {
                if (a.err && lane == 0) __hip_atomic_store(a.err, 12u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        asm volatile("" ::: "memory");
    };
    stage_load(0);
    uint32_t szcur = sz_at(0), sznext = sz_at(1);
    stage_store(0);
    group_sync(TN);
    TRACE_RT(2);

    for (int j = 0; j < nchunks; ++j) {
        const bool more = j + 1 < nchunks;
        if (more) stage_load(j + 1);
        const uint32_t szfar = sz_at(j + 2);
        // both steps of this chunk have landed once the second one has (DMAs land in issue order)
        const int need = (LD_DBG & 8) ? 0 : (2 * j + 1) * NC + c + 1;
        for (unsigned spins = 0; __builtin_amdgcn_readfirstlane(*landed) < need; ++spins) {
            __builtin_amdgcn_s_sleep(1);
            if (spins > LD_SPIN_LIMIT) {
                if (a.err && lane == 0) __hip_atomic_store(a.err, 13u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        asm volatile("" ::: "memory");
        if (j == 0) TRACE_RT(3);
        if (j == nchunks - 1) TRACE_RT(9);
        const u32x4 w0 = *reinterpret_cast<const u32x4*>(myring + ((2 * j) & (D - 1)) * 1024);
        const u32x4 w1 = *reinterpret_cast<const u32x4*>(myring + ((2 * j + 1) & (D - 1)) * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) consumed[c] = 2 * j + 2;
        const f16* xbuf = xs + (j & 1) * (32 * RSL) + xoff;
        const f16x2 szh = __builtin_bit_cast(f16x2, szcur);
        const f16 bA = szh[1] * (f16)-16.f;  // -16 (1024 + z + 1), exact
        const f16 bB = bA + (f16)15360.f;     // -16 (64 + z + 1), exact
        const f16x8 bp = {bA, bA, bA, bB, bB, bB, (f16)0.f, (f16)0.f};
        const f16x8 ap = ld16<f16x8>(apmine + (j & 1) * apstride);
        f32x16 g = mfma32(ap, bp, zero16);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const u32x4 cur = s2 ? w1 : w0;
            const f16* xk = xbuf + s2 * 64;
            f16x8 b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = unpack8(cur[i], EXA, EXB, M0r, M1r);
#pragma unroll
            for (int i = 0; i < 4; ++i) g = mfma32(ld16<f16x8>(xk + i * 8), b[i], g);
        }
        const float sc = (float)szh[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(sc, g[r], acc[r]);
        szcur = sznext;
        sznext = szfar;
        if (more) {
            stage_store((j + 1) & 1);
            group_sync(TN * (j + 2));
        }
        if (j == 0) TRACE_RT(4);
    }
    TRACE_RT(5);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // (2)
    TRACE_RT(6);

    // ---- sum the WK k-parts through LDS (fixed order => deterministic); the ring is free now ------------------------
    if (WK > 1) {
        float* red = reinterpret_cast<float*>(smem);  // [WK][TN tiles][64 lanes][16] <= 60 KiB
        if (wk > 0) {
            float* dst = red + (((wk * TN + wn) * 64 + lane) << 4);
#pragma unroll
            for (int r = 0; r < 16; r += 4)
                *reinterpret_cast<f32x4*>(dst + r) = f32x4{acc[r], acc[r + 1], acc[r + 2], acc[r + 3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // (3)
        TRACE_RT(7);
        if (wk > 0) return;
#pragma unroll
        for (int k2 = 1; k2 < WK; ++k2) {
            const float* src = red + (((k2 * TN + wn) * 64 + lane) << 4);
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(src + r);
                acc[r] += t[0];
                acc[r + 1] += t[1];
                acc[r + 2] += t[2];
                acc[r + 3] += t[3];
            }
        }
    }
    lean_epilogue<ACT>(la, acc, nt_raw, nt, split, mrows, lane);
    TRACE_RT(8);
}

}  // namespace gptq
