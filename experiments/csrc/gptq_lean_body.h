// "Lean" int4 GPTQ streaming GEMM for decode batches of up to 32 rows (round 3): the same block structure as
// gptq_gemm_unit (TN column tiles x WK k-parts, one wave each; x chunks double-buffered through LDS; weights streamed
// 1 KiB per wave-step through a register ring), with the dequantisation moved out of the VALU.
//
// Old step (per 1 KiB of weights = 2048 weights): 52 VALU to form f16((q - z) * s) + 4 MFMA  -> VALU-issue bound (73 VALU,
// 142 ns per wave-step per SIMD measured, tools/floor/lean.hip).  Lean step: the nibbles go to the MFMA as the f16 values
// 1024 + q (low nibble under exponent 0x6400) and 64 + q (high nibble under 0x5400): one v_and_or per pair, 20 VALU per
// step.  A per-group (128 rows) accumulator g takes the 8 MFMAs of the group plus ONE correction MFMA
//     g += A' . B',   A'[m] = split3(XA_g[m] / 16), split3(XB_g[m] / 16),  B'[n] = -16 (1024 + z + 1) x3, -16 (64 + z + 1) x3
// where XA_g[m] / XB_g[m] are the sums of x[m, k] over the group's rows that sit in low / high nibble positions
// (k % 4 < 2 / >= 2).  The offsets and the zero point cancel exactly (all factors are exact in f16, the sums are carried
// as three f16 terms = 33 bits), so g = sum_k x[m,k] (q[k,n] - (z+1)) up to fp32 accumulation, and the scale is applied
// once per group when g is folded into the output accumulator: out += s * g (16 FMAs per group).  Per step: ~31 VALU +
// 4.5 MFMA, 90 ns per wave-step per SIMD (MFMA-bound).  The weights are never rounded to f16 ((q - z) * s is exact in
// fp32 here; exllamav2 rounds it to f16 once) — results differ from the old kernel by that rounding only.
//
// The row sums of x come from the PRODUCER of x (rmsnorm, the gate_up epilogue of this kernel, tgis_xsum_f16): one pair
// {XA, XB} of fp32 per row and 16 columns, `xs[M][K/16][2]`.  Each k-part sums the 8 pairs of a group while it stages the
// x chunk, splits them and leaves the A' fragment in LDS next to the chunk.
//
// Preconditions (host, lean_ok): M <= 32, group size 128, K % 128 == 0, no act-order permutation, act 0 or 2.
#pragma once
#include "gptq_gemm_body.h"

namespace gptq {

struct LeanArgs {
    GemmArgs g;
    const float* xs;   // [M][ldxs][2] fp32: {XA, XB} of x per row and 16 k
    int64_t ldxs;      // 16-k blocks per row of xs (>= K / 16)
    float* xs_out;     // act == 2: the same sums of the activated output [M][N/32][2] (nullptr: not wanted)
};

constexpr int LEAN_XBYTES = 2 * 32 * RS * (int)sizeof(f16);  // x chunk double buffer of one k-part
constexpr int LEAN_APBYTES = 2 * 2 * 32 * 16;                // A' fragments: [buf][group of the chunk][row][8 halves]
__host__ __device__ constexpr int lean_lds_bytes(int WK) { return WK * (LEAN_XBYTES + LEAN_APBYTES) + 64; }

__device__ __forceinline__ float dpp_quad_swap1(float v) {  // lane i <- lane i ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_swap2(float v) {  // lane i <- lane i ^ 2
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_row_shr(float v, int n) {  // lane i <- lane i - n of the same 16-lane row, else 0
    if (n == 4)
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));
}

// v -> three f16 terms whose sum is v to 33 bits
__device__ __forceinline__ void split3(float v, f16& h, f16& m, f16& l) {
    h = (f16)v;
    const float r = v - (float)h;
    m = (f16)r;
    l = (f16)(r - (float)m);
}

// 8 nibbles -> (1024+n0, 1024+n4 | 64+n1, 64+n5 | 1024+n2, 1024+n6 | 64+n3, 64+n7): with the prepared nibble order
// [k0,k2,k4,k6,k1,k3,k5,k7] these are rows k0..k7 with offsets (1024, 1024, 64, 64, 1024, 1024, 64, 64)
__device__ __forceinline__ f16x8 unpack8(uint32_t q, uint32_t EXA, uint32_t EXB, uint32_t M0, uint32_t M1) {
    // plain C, not the asm and_or of gptq_gemm_body.h: these values feed the MFMA directly, and hipcc pads the
    // VALU-write -> MFMA-read hazard only for instructions it emitted itself (an asm and_or right in front of the MFMA
    // left rows 26/27/58/59 of every second step stale).  With the masks pinned to SGPRs and the exponents to VGPRs it
    // still selects one v_and_or_b32 per pair.
    const uint32_t q2 = q >> 8;
    u32x4 p = {(q & M0) | EXA, (q & M1) | EXB, (q2 & M0) | EXA, (q2 & M1) | EXB};
    return __builtin_bit_cast(f16x8, p);
}

// Epilogue shared by the lean units: lane holds out[m = (r&3) + 8 (r>>2) + 4 (lane>>5)][n = nt*32 + (lane&31)].
template <int ACT>
__device__ __forceinline__ void lean_epilogue(const LeanArgs& la, const f32x16& acc, const int nt_raw, const int nt,
                                              const int split, const int mrows, const int lane) {
    const GemmArgs& a = la.g;
    if (nt_raw >= a.NT) return;
    const int n = nt * 32 + (lane & 31);
    if (ACT == 2) {
        // interleaved gate/up image: lanes c < 16 hold gate column j = 16 nt + c, lanes c + 16 the matching up column;
        // out[m][j] = f16(f16(silu(f16 gate)) * f16 up) (flash_llama_modeling.py:332-335).  Host guarantees S == 1.
        const int c = lane & 31;
        const int half = a.N >> 1;
        const int j = nt * 16 + (c & 15);
        const int nsrc = (c < 16) ? j : half + j;
        const float bv = (a.bias && j < half) ? (float)a.bias[nsrc] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float mine = (float)(f16)(acc[r] + bv);
            const float other = __shfl_xor(mine, 16, 64);
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float sl = mine / (1.f + __expf(-mine));
            const f16 o = (f16)((float)(f16)sl * other);
            if (c < 16 && j < half && m < mrows) a.out[(int64_t)m * a.ldo + j] = o;
            if (la.xs_out) {
                // row sums of this tile's 16 activated columns for the consumer GEMM: lanes c % 4 < 2 carry low-nibble
                // rows of its image, c % 4 >= 2 high-nibble rows.  Pairs, then the four quads of the 16-lane row:
                // lanes 12/13 end with XA, lanes 14/15 with XB (fixed order).
                float v = (c < 16 && j < half) ? (float)o : 0.f;
                v += dpp_quad_swap1(v);
                v += dpp_row_shr(v, 4);
                v += dpp_row_shr(v, 8);
                if ((c == 12 || c == 14) && m < mrows) la.xs_out[((int64_t)m * a.NT + nt) * 2 + ((c >> 1) & 1)] = v;
            }
        }
        return;
    }
    if (a.S == 1 && !a.partial) {
        const float bv = (a.bias && n < a.N) ? (float)a.bias[n] : 0.f;
        if (n < a.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < mrows) a.out[(int64_t)m * a.ldo + n] = (f16)(acc[r] + bv);
            }
        }
    } else {
        float* sl = a.slabs + ((int64_t)split * 32) * (a.NT * 32) + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            sl[(int64_t)m * (a.NT * 32)] = acc[r];
        }
    }
}

template <int TN, int WK, int ACT, int RING>
__device__ __forceinline__ void gptq_lean_unit(const LeanArgs& la, const int ntg, const int split,
                                               unsigned char* smem) {
    static_assert(RING == 4 || RING == 8, "one or two chunks of weights in flight");
    static_assert(ACT == 0 || ACT == 2, "plain or SiLU(gate) * up epilogue");
    const GemmArgs& a = la.g;
    constexpr int NWAVES = TN * WK;
    constexpr int GT = 64 * TN;                     // threads of one k-part group
    constexpr int NJ = (32 * 32 + GT - 1) / GT;     // 16-byte x pieces per thread per chunk
    constexpr int RSTEP = GT / 32;                  // rows covered by one pass of the group
    constexpr int NQ = (256 + GT - 1) / GT;         // quarter-tasks of the row-sum staging per thread per chunk
    constexpr int RG = RING / 2;                    // groups in the ring
    const int tid = threadIdx.x, lane = tid & 63;
    TRACE(0);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w % TN, wk = w / TN, ltid = wn * 64 + lane;
    f16* xs = reinterpret_cast<f16*>(smem + wk * LEAN_XBYTES);                         // [2][32][RS]
    unsigned char* aps = smem + WK * LEAN_XBYTES + wk * LEAN_APBYTES;                  // [2][2][32][16 B]
    unsigned char* ctrl = smem + WK * (LEAN_XBYTES + LEAN_APBYTES);                    // 16 B of zeros, then counters
    const int mrows = min(32, a.M);
    const int krp = a.KR / WK;
    const int k0 = split * a.KR + wk * krp;
    const int k1 = min(a.K, k0 + krp);
    const int nchunks = krp / KC;
    const int nt_raw = ntg * TN + wn;
    const int nt = min(nt_raw, a.NT - 1);
    const int ks0 = k0 >> 6;
    const int ks_last = a.KS - 2;

    const char* wtile = reinterpret_cast<const char*>(a.prep) + (int64_t)nt * a.KS * 1024;
    const char* sztile = reinterpret_cast<const char*>(a.prep + a.offB) + (int64_t)nt * a.G * 128;
    const uint32_t woff = lane * 16, szoff = (lane & 31) * 4;
    const int ks_clamp = min(ks_last, max(ks0, ((k1 + 63) >> 6) - 1));
    const int g0 = k0 >> 7;                 // first group of this wave's rows
    const int g_end = k1 >> 7;              // first group past them: its {scale, zero} word is forced to zero
    auto sz_at = [&](int grp) -> uint32_t {  // one {scale, 1024 + z + 1} pair per lane and 128-row group
        const char* p = sztile + (int64_t)min(g0 + grp, a.G - 1) * 128;
        PIN_SGPR(p);
        const uint32_t v = *(const GLOBAL_AS uint32_t*)(p + szoff);
        return g0 + grp < g_end ? v : 0u;
    };
    auto w_at = [&](int step) -> u32x4 {
        const char* p = wtile + (int64_t)min(ks0 + step, ks_clamp) * 1024;
        PIN_SGPR(p);
        return __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(p + woff));
    };
    u32x4 wq[RING];
    uint32_t szg[RG];

    // ---- x staging (as gptq_gemm_unit) + row-sum staging ------------------------------------------------------------
    const f16* xbase = a.x;
    const int srow = ltid >> 5, scol = (ltid & 31) * 8;
    f16x8 xg[NJ] = {};
    uint32_t rowoff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rowoff[j] = (uint32_t)(min(srow + RSTEP * j, mrows - 1) * (int)a.ldx * 2);
    // quarter-task q: row q >> 3, group (q >> 2) & 1 of the chunk, quarter q & 3 = two of the group's eight 16-k blocks
    f32x4 xq[NQ] = {};
    uint32_t xsrow[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = ltid + i * GT;
        xsrow[i] = (uint32_t)(min(q >> 3, mrows - 1) * (int)la.ldxs);
    }
    const int kb_last = (a.K >> 4) - 2;
    auto stage_load = [&](int chunk) {
        const int kc = min(k0 + chunk * KC + scol, a.K - 8);
        const char* xb = reinterpret_cast<const char*>(xbase);
        PIN_SGPR(xb);
#if !defined(LEAN_ABL) || !(LEAN_ABL & 4)
#pragma unroll
        for (int j = 0; j < NJ; ++j) xg[j] = *(const GLOBAL_AS f16x8*)(xb + rowoff[j] + (uint32_t)kc * 2);
#endif
        const char* sb = reinterpret_cast<const char*>(la.xs);
        PIN_SGPR(sb);
        const int kb0 = (k0 + chunk * KC) >> 4;
#if !defined(LEAN_ABL) || !(LEAN_ABL & 2)
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = ltid + i * GT;
            const int kb = min(kb0 + ((q >> 2) & 1) * 8 + (q & 3) * 2, kb_last);
            xq[i] = *(const GLOBAL_AS f32x4*)(sb + ((size_t)xsrow[i] + (uint32_t)kb) * 8);
        }
#endif
    };
    auto stage_store = [&](int buf) {
        f16* dst = xs + buf * (32 * RS) + srow * RS + scol;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (NJ * RSTEP == 32 || srow + RSTEP * j < 32) st16(dst + j * RSTEP * RS, xg[j]);
#if !defined(LEAN_ABL) || !(LEAN_ABL & 2)
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = ltid + i * GT;
            float xa = xq[i][0] + xq[i][2], xb = xq[i][1] + xq[i][3];
            xa += dpp_quad_swap1(xa);
            xb += dpp_quad_swap1(xb);
            xa += dpp_quad_swap2(xa);
            xb += dpp_quad_swap2(xb);
            f16 ah, am, al, bh, bm, bl;
            split3(xa * 0.0625f, ah, am, al);
            split3(xb * 0.0625f, bh, bm, bl);
            const f16x8 ap = {ah, am, al, bh, bm, bl, (f16)0.f, (f16)0.f};
            if ((q & 3) == 0 && (NQ * GT == 256 || q < 256))
                st16(aps + ((buf * 2 + ((q >> 2) & 1)) * 32 + (q >> 3)) * 16, ap);
        }
#endif
    };

    uint32_t EXA = 0x64006400u, EXB = 0x54005400u, M0r = 0x000F000Fu, M1r = 0x00F000F0u;
    asm volatile("" : "+v"(EXA), "+v"(EXB));
    asm volatile("" : "+s"(M0r), "+s"(M1r));
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc = zero16;
    const int xoff = (lane & 31) * RS + (lane >> 5) * 32;
    // A' fragment address of this lane: rows sit in lanes 0..31 (k slots 0..7); lanes 32..63 (k slots 8..15) read zeros
    const uint32_t apoff = lane < 32 ? (uint32_t)(WK * LEAN_XBYTES + wk * LEAN_APBYTES + lane * 16)
                                     : (uint32_t)(WK * (LEAN_XBYTES + LEAN_APBYTES));

    volatile lds_int* sync_cnt = (volatile lds_int*)(ctrl + 16) + wk;
    if (tid < 4) reinterpret_cast<uint32_t*>(ctrl)[tid] = 0u;
    if (wn == 0 && lane == 0) *sync_cnt = 0;
    stage_load(0);
#pragma unroll
    for (int s = 0; s < RG; ++s) szg[s] = sz_at(s);
#pragma unroll
    for (int s = 0; s < RING; ++s) wq[s] = w_at(s);
    TRACE(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto group_sync = [&](int target) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lds_int*)sync_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__builtin_amdgcn_readfirstlane(*sync_cnt) < target) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    stage_store(0);
    group_sync(TN);
    TRACE(2);

    // One chunk = 4 k64-steps = 2 groups.  SB: first ring slot of the chunk; STAGE: a next chunk exists; REFILL: the chunk
    // RING / 4 ahead exists (its weights and scales replace this chunk's in place).
    auto chunk_body = [&](const int chunk, auto sb_tag, auto stage_tag, auto refill_tag) {
        constexpr int SB = decltype(sb_tag)::value;
        constexpr bool STAGE = decltype(stage_tag)::value, REFILL = decltype(refill_tag)::value;
        if (STAGE) stage_load(chunk + 1);
        uint32_t szn[2];
        if (REFILL) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) szn[gi] = sz_at(chunk * 2 + gi + RG);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f16* xbuf = xs + (chunk & 1) * (32 * RS) + xoff;
        const unsigned char* apbuf = smem + apoff + (lane < 32 ? (chunk & 1) * (2 * 32 * 16) : 0);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const f16x2 szh = __builtin_bit_cast(f16x2, szg[SB / 2 + gi]);
            const f16 bA = szh[1] * (f16)-16.f;        // -16 (1024 + z + 1), exact
            const f16 bB = bA + (f16)15360.f;           // -16 (64 + z + 1), exact
            const f16x8 bp = {bA, bA, bA, bB, bB, bB, (f16)0.f, (f16)0.f};
#if defined(LEAN_ABL) && (LEAN_ABL & 1)
            f32x16 g = acc;
#else
            const f16x8 ap = ld16<f16x8>(apbuf + (lane < 32 ? gi * (32 * 16) : 0));
            f32x16 g = mfma32(ap, bp, zero16);
#endif
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int s4 = gi * 2 + s2;
                const int step = chunk * 4 + s4;
                const u32x4 cur = wq[SB + s4];
                const f16* xk = xbuf + s4 * 64;
                f16x8 b[4];
#if defined(LEAN_ABL) && (LEAN_ABL & 32)
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(f16x8, u32x4{cur[i], cur[i], cur[i], cur[i]});
#else
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = unpack8(cur[i], EXA, EXB, M0r, M1r);
#endif
#if !defined(LEAN_ABL) || !(LEAN_ABL & 16)
                if (REFILL) {
                    __builtin_amdgcn_sched_barrier(0);
                    wq[SB + s4] = w_at(step + RING);
                    __builtin_amdgcn_sched_barrier(0);
                }
#else
                wq[SB + s4][0] += step;
#endif
#if defined(LEAN_ABL) && (LEAN_ABL & 32)
                g[0] += (float)(b[0][0] + b[1][1] + b[2][2] + b[3][3]);
#else
#pragma unroll
                for (int i = 0; i < 4; ++i) g = mfma32(ld16<f16x8>(xk + i * 8), b[i], g);
#endif
            }
#if defined(LEAN_ABL) && (LEAN_ABL & 1)
            acc = g;
            acc[0] += (float)szh[0] + (float)bp[0];
#else
            const float sc = (float)szh[0];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(sc, g[r], acc[r]);
#endif
        }
        if (REFILL) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) szg[SB / 2 + gi] = szn[gi];
        }
        if (!STAGE) return;
#if !defined(LEAN_ABL) || !(LEAN_ABL & 8)
        stage_store((chunk + 1) & 1);
        TRACE(3 + 2 * min(chunk, 3));
        group_sync(TN * (chunk + 2));
        TRACE(4 + 2 * min(chunk, 3));
#endif
    };
    using I0 = std::integral_constant<int, 0>;
    using I4 = std::integral_constant<int, 4>;
    using Y = std::true_type;
    using N = std::false_type;
    if (RING == 4) {
        for (int chunk = 0; chunk + 1 < nchunks; ++chunk) chunk_body(chunk, I0{}, Y{}, Y{});
        chunk_body(nchunks - 1, I0{}, N{}, N{});
    } else {
        int chunk = 0;
        for (; chunk + 3 < nchunks; chunk += 2) {
            chunk_body(chunk, I0{}, Y{}, Y{});
            chunk_body(chunk + 1, I4{}, Y{}, Y{});
        }
        const int left = nchunks - chunk;  // 1, 2 or 3 (wave-uniform)
        if (left == 3) {
            chunk_body(chunk, I0{}, Y{}, Y{});
            chunk_body(chunk + 1, I4{}, Y{}, N{});
            chunk_body(chunk + 2, I0{}, N{}, N{});
        } else if (left == 2) {
            chunk_body(chunk, I0{}, Y{}, N{});
            chunk_body(chunk + 1, I4{}, N{}, N{});
        } else {
            chunk_body(chunk, I0{}, N{}, N{});
        }
    }
    TRACE(9);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every k-part is done with its x buffers: the reduction below reuses them

    // ---- sum the WK k-parts through LDS (fixed order => deterministic) ------------------------------------------------
    if (WK > 1) {
        float* red = reinterpret_cast<float*>(smem);  // [WK][TN tiles][64 lanes][16]
        if (wk > 0) {
            float* dst = red + (((wk * TN + wn) * 64 + lane) << 4);
#pragma unroll
            for (int r = 0; r < 16; r += 4)
                *reinterpret_cast<f32x4*>(dst + r) = f32x4{acc[r], acc[r + 1], acc[r + 2], acc[r + 3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        TRACE(11);
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

    TRACE(12);
    lean_epilogue<ACT>(la, acc, nt_raw, nt, split, mrows, lane);
    TRACE(13);
}

// Whether the lean kernel covers this GEMM (otherwise the caller keeps gptq_gemm_kernel).
static inline bool lean_ok(int64_t M, int64_t K, int64_t N, int64_t groups, bool act_order, int act) {
    if (M < 1 || M > 32 || act_order || (act != 0 && act != 2)) return false;
    if (groups <= 0 || K % groups != 0 || K / groups != 128 || K % 128 != 0 || N % 32 != 0) return false;
    return true;
}

}  // namespace gptq
