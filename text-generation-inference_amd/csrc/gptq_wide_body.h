// The int4 GPTQ GEMM for decode batches of up to 32 rows whose activation arrives in MFMA-fragment order (round 4).
//
// Replaces exllamav2_kernels.gemm_half_q_half (utils/gptq/exllamav2.py:139-144) on the decode step, like the streaming
// kernel of gptq_gemm_body.h, on the same prepared image.  What is different, and why (tools/floor/wide.hip measured every
// step on the cfg3 shapes, profiles/r04_wide_*.log):
//   * the activation is not staged through LDS chunk by chunk.  Its producer (add + RMSNorm, the attention epilogue, the
//     SiLU * up epilogue of this kernel) writes it in FRAGMENT ORDER — xf[k64-step][i][lane][8 halves], lane = 32 (k / 32 % 2)
//     + row, i = k / 8 % 4 (xf_off in common.h) — so the A operand of the four MFMAs of a k64-step is four contiguous
//     one-KiB loads straight into registers, and a wave needs nobody else's data until the very end: no chunk hand-offs,
//     no arrival counters, no barrier in front of the first MFMA.  Row-major x read the same way costs 1.4 - 2.6 us per
//     launch (32 rows at a stride of 8 KiB: 32 cache lines per load instruction, all in one L1 set);
//   * a wave owns CT (2 - 4) 32-column tiles, not one: every A fragment feeds CT MFMAs;
//   * the 8 waves of a block split the block's k range; their fp32 sums meet once, in LDS laid out [k-part][tile][register]
//     [lane] — every access 64 consecutive words (the [lane][16] layout of the streaming kernel is a 16-way bank conflict:
//     2.3 us of a 10 us launch) — and every wave finishes two of the sixteen accumulator rows of each tile (fixed order);
//   * two k64-steps of weights + scales + activation in flight per wave, refilled in place; the last steps are taken by
//     wave-uniform branches, so no wave dequantises padding.
// cfg3 shapes, 32 rows, us per launch (streaming kernel -> this one): qkv + rope epilogue 16.9 -> 12.x, o 6.0 -> 5.3,
// gate_up + SiLU 16.8 -> 14.x, down 10.5 -> 9.2.
#pragma once
#include "gptq_gemm_body.h"

// tools/floor/wide_unit.hip compiles this unit with WIDE_STAMP defined (s_memtime stamps per wave); the library does not.
#ifndef WIDE_STAMP
#define WIDE_STAMP(i)
#endif

namespace gptq {

constexpr int WIDE_WK = 8;      // k-parts (waves) per block
constexpr int WIDE_DEPTH = 2;   // k64-steps in flight per wave
constexpr int WIDE_TG(int ct) { return ct < 4 ? ct : 4; }   // tiles per round of the k-part exchange
static inline size_t wide_lds_bytes(int ct) { return (size_t)WIDE_WK * WIDE_TG(ct) * 4096; }

struct WidePlan {
    int CT, S;  // column tiles per wave (= per block), global k splits
};

// Does the fragment-order kernel serve this GEMM at all?  (<= 64 rows, whole k64-steps, one {scale, zero} per k64-step and
// column, no act-order permutation.)
static inline bool wide_serves(int64_t M, int64_t K, int64_t N, int64_t groups, bool act_order) {
    if (M < 1 || M > 64 || act_order || K % 64 || N % 32 || groups < 1 || K % groups) return false;
    const int64_t gs = K / groups;
    if (groups == 1) return true;
    if (gs % 64) return false;
    const int64_t spg = gs / 64;
    return (spg & (spg - 1)) == 0;
}

// One block per CU where the shape allows it: column groups first (a wider group re-uses an A fragment more often), then
// global k splits (fp32 slabs that the consumer sums) until ~256 blocks; at least one k64-step per wave.
// act 2 / 3 (SiLU * up, rotary + cache write) finish their outputs in the epilogue: no split.
// Round 5 (tensor-parallel shard shapes): CT 1 where two tiles per wave would leave fewer than 128 blocks on an unsplit plan
// (a shard's gate_up: 224 tiles = 112 blocks of two tiles, each taking in its whole M x K activation — 1 MB at 64 rows —
// through one CU), and `finished` (the caller needs the f16 output, not slabs: row-parallel linears in front of an
// all-reduce): unsplit whenever that still gives >= 128 blocks, because a split plan costs a reduce launch on top.
static inline WidePlan plan_wide(int64_t K, int64_t N, int act, int64_t M = 32, bool finished = false) {
    if (const char* ov = getenv("TGIS_GPTQ_WIDE_PLAN")) {  // tuning hook: "CT,S"
        int ct = 0, sp = 0;
        if (sscanf(ov, "%d,%d", &ct, &sp) == 2 && ct >= 1 && ct <= 4 && sp >= 1 && (sp == 1 || (act != 2 && act != 3)))
            return {ct, sp};
    }
    const int64_t tiles = cdiv64(N, 32), steps = K / 64;
    int CT = 2;
    while (CT < 4 && cdiv64(tiles, CT) > 256) ++CT;
    const bool unsplit = act == 2 || act == 3;
    // (SiLU * up of a 64-row shard with a long k range — 70B gate_up at TP = 8, 224 one-tile blocks that each take in a 1 MiB
    // activation, 20.3 us for 29 MB — was also run split over k with the activation in the reduce launch, CT x S = 1x2, 2x2,
    // 4x4, 4x2, 2x4: none beat the unsplit plan on the rank step, profiles/r05_tp8_silu_plans.log.)
    if (tiles <= 256 && (unsplit ? cdiv64(tiles, 2) < 128 : (finished && tiles >= 128))) return {1, 1};
    if (finished && !unsplit && cdiv64(tiles, CT) >= 128) return {CT, 1};
    int64_t S = 1;
    if (!unsplit) {
        const int64_t cgs = cdiv64(tiles, CT);
        S = std::max<int64_t>(1, (256 + cgs / 2) / cgs);
        S = std::min<int64_t>(S, std::max<int64_t>(1, steps / WIDE_WK));
        while (S > 1 && (S - 1) * cdiv64(steps, S) >= steps) --S;  // no empty last split
        // 64 rows: the activation is 8 KiB per k64-step against CT KiB of weights — with a long k range four tiles per wave
        // and twice the splits move half the activation bytes per weight byte (70B down projection 47 -> 41 us; shorter
        // ranges lose more to the extra slabs: 70B o 17.4 vs 18.9, 7B down 14.2 vs 15.5 — tools/floor/wide_unit.hip)
        if (M > 32 && CT == 2 && tiles % 4 == 0 && steps / (WIDE_WK * 2 * S) >= 12) {
            CT = 4;
            S *= 2;
        }
    }
    return {CT, (int)S};
}
static inline int64_t wide_blocks(int64_t K, int64_t N, int act, int64_t M = 32, bool finished = false) {
    const WidePlan p = plan_wide(K, N, act, M, finished);
    return cdiv64(cdiv64(N, 32), p.CT) * p.S;
}
// the largest split count either row class (<= 32, <= 64) may use: what slab buffers are sized for
static inline int wide_max_splits(int64_t K, int64_t N) {
    return std::max(std::max(plan_wide(K, N, 0, 32).S, plan_wide(K, N, 0, 64).S),
                    std::max(plan_wide(K, N, 0, 32, true).S, plan_wide(K, N, 0, 64, true).S));
}

// OUTF: the act = 2 output (the operand of the down projection) leaves in fragment order as well.
// MR = 32-row blocks of the activation (1: M <= 32; 2: M <= 64 — every dequantised B fragment then feeds two MFMAs, the
// "two tiles per wave on one activation" of the 64-row passes, with the activation in registers instead of LDS).
template <int CT, int ACT, bool OUTF, int MR>
__device__ __forceinline__ void gptq_wide_unit(const GemmArgs& a, unsigned char* smem) {
    constexpr int WK = WIDE_WK, DEPTH = WIDE_DEPTH, NR = 16 / WK;
    {   // every cache line of the argument block is requested at entry: one scalar round trip instead of three dependent
        // ones before the first weight request (tools/floor/wide.hip `pre`: 0.1 - 0.25 us per launch)
        const int64_t l0 = a.ldo;
        const int l1 = a.S, l2 = a.rD;
        asm volatile("" ::"s"(l0), "s"(l1), "s"(l2));
    }
    const int lane = threadIdx.x & 63;
    const int wk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cg = blockIdx.x, split = blockIdx.y;
    const int steps = a.K >> 6;
    const int sp_len = (steps + a.S - 1) / a.S;
    const int sb = split * sp_len, se = min(steps, sb + sp_len);
    const int len = max(se - sb, 0);
    // This wave's k64-steps (may be empty).  (Round 5 measured an uneven split — the four waves a CU launches first run ~2 steps
    // ahead of the younger four, which queue behind them in the address path — and a prologue barrier, a raised priority for
    // the younger half and deeper weight rings: the block finishes when its CU has moved its bytes, however they are dealt;
    // tools/floor/wide.hip `kb` / `pb` / `prio` / `DW`, profiles/r05_wide_kbias*.log, r05_wide_plans.log.)
    const int s0 = sb + (len * wk) / WK, s1 = sb + (len * (wk + 1)) / WK;
    const int mrows = a.M;  // 1 .. 32 MR

    const char* wt[CT];
    const char* st[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int nt = min(cg * CT + t, a.NT - 1);  // a tile past the matrix re-reads the last one and is never stored
        wt[t] = reinterpret_cast<const char*>(a.prep) + (int64_t)nt * a.KS * 1024;
        st[t] = reinterpret_cast<const char*>(a.prep + a.offB) + (int64_t)nt * a.G * 128;
    }
    const uint32_t woff = lane * 16, szoff = (lane & 31) * 4;
    const int sclamp = max(s1 - 1, s0);  // loads past the wave's steps re-read its last one (a cache hit), never consumed
    const char* xb = reinterpret_cast<const char*>(a.x);
    const int64_t xblk = (int64_t)a.K * 64;  // bytes between the 32-row blocks of the activation

    u32x4 wq[DEPTH][CT];
    uint32_t sz[DEPTH][CT];
    f16x8 xa[DEPTH][MR][4];
    auto load_step = [&](int d, int step) {
        const int sc = min(step, sclamp);
        const int g = min(sc >> a.spg_shift, a.G - 1);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const char* p = st[t] + (int64_t)g * 128;
            PIN_SGPR(p);
            sz[d][t] = *(const GLOBAL_AS uint32_t*)(p + szoff);
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const char* p = wt[t] + (int64_t)sc * 1024;
            PIN_SGPR(p);
            wq[d][t] = __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(p + woff));
        }
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const char* p = xb + mr * xblk + (int64_t)sc * 4096;  // fragment order: four contiguous KiB per k64-step
            PIN_SGPR(p);
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[d][mr][i] = *(const GLOBAL_AS f16x8*)(p + woff + i * 1024);
        }
    };

    // the rows this wave finishes: accumulator registers [wk NR, (wk + 1) NR) of every tile and row block
    auto row_of = [&](int mr, int j) {
        const int r = wk * NR + j;
        return mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    };
    // ACT 3: their cache slots and rotary positions
    int32_t rpos[ACT == 3 ? MR : 1][ACT == 3 ? NR : 1], rslot[ACT == 3 ? MR : 1][ACT == 3 ? NR : 1];
    if (ACT == 3) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int m = min(row_of(mr, j), mrows - 1);
                rpos[mr][j] = a.positions[m];
                rslot[mr][j] = a.slots[m];
            }
    }

    uint32_t EXr = 0x64006400u, M0r = 0x000F000Fu, M1r = 0x00F000F0u;
    asm volatile("" : "+v"(EXr));
    asm volatile("" : "+s"(M0r), "+s"(M1r));
    f32x16 acc[MR][CT];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[mr][t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    WIDE_STAMP(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_step(d, s0 + d);
    WIDE_STAMP(1);

    auto consume = [&](int d) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const f16x2 szh = __builtin_bit_cast(f16x2, sz[d][t]);
            const f16 zc1 = szh[1];
            const f16 zd1 = (f16)960.f - zc1;  // -(64 + z + 1), exact
            const f16x2 zc = {zc1, zc1}, zd = {zd1, zd1}, sc = {szh[0], szh[0]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 b = dequant8(wq[d][t][i], zc, zd, sc, EXr, M0r, M1r);
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) acc[mr][t] = mfma32(xa[d][mr][i], b, acc[mr][t]);
            }
        }
    };

    int s = s0;
    for (; s + DEPTH < s1; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            consume(d);
            __builtin_amdgcn_sched_barrier(0);
            load_step(d, s + d + DEPTH);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    WIDE_STAMP(2);
    // the last group: only the steps that exist (wave-uniform branches; nothing is requested any more)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (s + d < s1) consume(d);
    WIDE_STAMP(3);

    // ACT 3: the finishing rows' cos / sin entries are requested before the exchange (their positions came in at entry)
    f16 rcos[ACT == 3 ? MR : 1][ACT == 3 ? CT : 1][ACT == 3 ? NR : 1], rsin[ACT == 3 ? MR : 1][ACT == 3 ? CT : 1][ACT == 3 ? NR : 1];
    if (ACT == 3) {
        const int per = a.rD >> 5;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int nt = min(cg * CT + t, a.NT - 1);
            const int tt = nt - (nt / per) * per;
            const int dr = 16 * tt + (lane & 15);
            const bool roth = nt / per < a.rH + a.rHkv;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    rcos[mr][t][j] = roth ? a.cosb[(int64_t)rpos[mr][j] * (a.rD >> 1) + dr] : (f16)1.f;
                    rsin[mr][t][j] = roth ? a.sinb[(int64_t)rpos[mr][j] * (a.rD >> 1) + dr] : (f16)0.f;
                }
        }
    }

    // ---- k-part sum through LDS, one 32-row block at a time: [k-part][tile][register][lane] (every access 64 consecutive
    // words), then wave wk sums registers [wk NR, (wk + 1) NR) of every tile in the fixed order of the k-parts ----
    float* red = reinterpret_cast<float*>(smem);
    const int c = lane & 31;
    // (CT > 4: the tiles meet in groups of TG = 4, so that the scratch stays at WK x 4 x 4 KiB = 128 KiB)
    constexpr int TG = WIDE_TG(CT);
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
      for (int tg0 = 0; tg0 < CT; tg0 += TG) {
        if (mr > 0 || tg0 > 0) __syncthreads();  // the previous group's sums have been read
        // (round 5: register PAIRS — [k-part][tile][register / 2][lane][2] — so that the sixteen registers leave as eight
        // ds_write_b64 and the two registers a wave finishes come back as one ds_read_b64 per k-part and tile: the exchange
        // moves 2 x 8 waves x CT x 4 KiB through an LDS that takes 64 B/clk of 4-byte stores and gives 128 B/clk of 4-byte
        // loads, ~2.3 k clocks at CT 3; 8-byte accesses take 85 and 256 B/clk.  Same values, same order of the sum.)
        static_assert(NR == 2, "the pair layout is for eight k-parts");
#pragma unroll
        for (int t = tg0; t < tg0 + TG && t < CT; ++t) {
            f32x2* dst = reinterpret_cast<f32x2*>(red + ((wk * TG + (t - tg0)) << 10)) + lane;
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) dst[rp << 6] = f32x2{acc[mr][t][2 * rp], acc[mr][t][2 * rp + 1]};
        }
        __syncthreads();
        float fin[CT][NR];
#pragma unroll
        for (int t = tg0; t < tg0 + TG && t < CT; ++t) {
#pragma unroll
            for (int k2 = 0; k2 < WK; ++k2) {
                const f32x2 v = (reinterpret_cast<const f32x2*>(red + ((k2 * TG + (t - tg0)) << 10)) + (wk << 6))[lane];
#pragma unroll
                for (int j = 0; j < NR; ++j) fin[t][j] = k2 == 0 ? v[j] : fin[t][j] + v[j];
            }
        }
#pragma unroll
        for (int t = tg0; t < tg0 + TG && t < CT; ++t) {
            const int nt = cg * CT + t;
            if (nt >= a.NT) break;
            if (ACT == 3) {
                // rope image: see gptq_gemm_body.h (the same epilogue on the same image)
                const int per = a.rD >> 5;
                const int head = nt / per, tt = nt - head * per;
                const bool roth = head < a.rH + a.rHkv;
                const int d = roth ? ((c < 16) ? 16 * tt + c : (a.rD >> 1) + 16 * tt + (c - 16)) : 32 * tt + c;
                const int col = head * a.rD + d;
                const float bv = a.bias ? (float)a.bias[col] : 0.f;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int m = row_of(mr, j);
                    const float mine = (float)(f16)(fin[t][j] + bv);
                    float o = mine;
                    if (roth) {
                        const float other = __shfl_xor(mine, 16, 64);
                        const float cf = (float)rcos[mr][t][j], sf = (float)rsin[mr][t][j];
                        o = (c < 16) ? mine * cf - other * sf : other * sf + mine * cf;
                    }
                    const f16 oh = (f16)o;
                    if (m < mrows) {
                        if (head < a.rH) {
                            a.out[(int64_t)m * a.ldo + col] = oh;
                        } else {
                            const int page = rslot[mr][j] >> 5, tok = rslot[mr][j] & 31;
                            if (roth)
                                a.kpool[((int64_t)page * a.rHkv + (head - a.rH)) * 32 * a.rD + k_off(tok, d, a.rD)] = oh;
                            else
                                a.vpool[((int64_t)page * a.rHkv + (head - a.rH - a.rHkv)) * 32 * a.rD + v_off(tok, d, a.rD)] = oh;
                        }
                    }
                }
                continue;
            }
            const int n = nt * 32 + c;
            if (ACT == 2) {
                // interleaved gate / up image: lanes c < 16 hold gate column j2, lanes c + 16 the matching up column
                const int half = a.N >> 1;
                const int j2 = nt * 16 + (c & 15);
                const int nsrc = (c < 16) ? j2 : half + j2;
                const float bv = (a.bias && j2 < half) ? (float)a.bias[nsrc] : 0.f;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const float mine = (float)(f16)(fin[t][j] + bv);
                    const float other = __shfl_xor(mine, 16, 64);
                    const int m = row_of(mr, j);
                    if (c < 16 && j2 < half && m < mrows) {
                        const float sl = mine / (1.f + __expf(-mine));
                        const f16 o = (f16)((float)(f16)sl * other);
                        if (OUTF)
                            a.out[xf_off(m, j2, half)] = o;
                        else
                            a.out[(int64_t)m * a.ldo + j2] = o;
                    }
                }
                continue;
            }
            if (a.S == 1 && !a.partial) {
                const float bv = (a.bias && n < a.N) ? (float)a.bias[n] : 0.f;
                if (n < a.N) {
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        const int m = row_of(mr, j);
                        if (m < mrows) a.out[(int64_t)m * a.ldo + n] = (f16)(fin[t][j] + bv);
                    }
                }
            } else {
                // slabs in 32-row units: [row block][split][32][ld]
                float* sl = a.slabs + ((int64_t)(mr * a.S + split) * 32) * (a.NT * 32) + n;
#pragma unroll
                for (int j = 0; j < NR; ++j) sl[(int64_t)(row_of(mr, j) & 31) * (a.NT * 32)] = fin[t][j];
            }
        }
      }
    }
    WIDE_STAMP(4);
}

}  // namespace gptq
