// EXPERIMENT (round 5; experiments/README.md): measured ON PAR with the product's streaming dense kernel, so it is not part
// of libtgis_hip.so.  The full wiring (entry points, ctypes binding, FastLinear, tests) is commit 1e13390.
//
// The dense (f16 / bf16 weights) decode GEMM for batches of up to 32 rows whose activation arrives in MFMA-fragment order
// (round 5) — the structure of gptq_wide_body.h without the dequantisation.
//
// Replaces cuBLAS `F.linear` / `torch.mm` at decode M (utils/layers.py:110-111 of the reference) like the streaming kernel of
// dense_gemm_body.h, on the same prepared image ([NT][KS][4][64 lanes][8]: the four KiB of a tile's k64-step ARE the B
// operands of its four MFMAs).  What is different: the activation is not staged through LDS chunk by chunk — its producer
// (add + RMSNorm, the attention epilogue, the SiLU * up epilogue of this kernel) writes it in fragment order (xf_off in
// common.h), so the A operand of a k64-step is four contiguous one-KiB loads straight into registers; a wave owns CT 32-column
// tiles over its own k range (every A fragment feeds CT MFMAs), keeps two k64-steps of weights + activation in flight and
// meets the block's other seven k-parts once, in LDS (8-byte accesses, fixed order of the sum).
#pragma once
#include "dense_gemm_body.h"

namespace dense {

constexpr int DWIDE_WK = 8;     // k-parts (waves) per block
constexpr int DWIDE_DEPTH = 2;  // k64-steps in flight per wave
static inline size_t dwide_lds_bytes(int ct) { return (size_t)DWIDE_WK * ct * 4096; }

struct DWidePlan {
    int CT, S;  // column tiles per wave (= per block), global k splits
};

// Does the fragment-order kernel serve this GEMM at all?  (<= 32 rows, whole k64-steps)
static inline bool dwide_serves(int64_t M, int64_t K, int64_t N) { return M >= 1 && M <= 32 && K % 64 == 0 && N % 32 == 0; }

// As plan_wide (gptq_wide_body.h): column groups first, then global k splits until the grid covers the chip in ONE round;
// a tile-step is 4 KiB here (1 KiB there), so wide tiles are dearer in registers (16 VGPRs per tile-step in flight) and the
// best (CT, S) is searched: the plan with the most blocks <= 256, wider tiles on a tie (less activation traffic per weight
// byte), at least one k64-step per wave.  unsplit: the epilogue needs the finished sum (SiLU * up, rotary, GELU, or an f16
// output that feeds an all-reduce).
static inline DWidePlan plan_dwide(int64_t K, int64_t N, bool unsplit) {
    if (const char* ov = getenv("TGIS_DENSE_WIDE_PLAN")) {  // tuning hook: "CT,S"
        int ct = 0, sp = 0;
        if (sscanf(ov, "%d,%d", &ct, &sp) == 2 && ct >= 1 && ct <= 4 && sp >= 1 && (sp == 1 || !unsplit)) return {ct, sp};
    }
    const int64_t tiles = cdiv64(N, 32), steps = K / 64;
    DWidePlan best{1, 1};
    int64_t best_blocks = -1;
    for (int ct = 1; ct <= 4; ++ct) {
        const int64_t cgs = cdiv64(tiles, ct);
        if (cgs > 256 && ct < 4) continue;  // more than one round: take wider tiles
        int64_t S = unsplit ? 1 : std::max<int64_t>(1, 256 / cgs);
        S = std::min<int64_t>(S, std::max<int64_t>(1, steps / DWIDE_WK));
        while (S > 1 && (S - 1) * cdiv64(steps, S) >= steps) --S;  // no empty last split
        const int64_t blocks = std::min<int64_t>(cgs * S, 256);
        if (blocks > best_blocks || (blocks == best_blocks && ct > best.CT && blocks >= 224)) {
            best_blocks = blocks;
            best = {ct, (int)S};
        }
    }
    return best;
}
static inline int64_t dwide_blocks(int64_t K, int64_t N, bool unsplit) {
    const DWidePlan p = plan_dwide(K, N, unsplit);
    return cdiv64(cdiv64(N, 32), p.CT) * p.S;
}

// OUTF: the act = 2 output (the operand of the down projection) leaves in fragment order as well.
template <typename T, int CT, int ACT, bool OUTF>
__device__ __forceinline__ void dense_wide_unit(const DenseArgs& a, unsigned char* smem) {
    using V8 = typename VecT<T>::x8;
    constexpr int WK = DWIDE_WK, DEPTH = DWIDE_DEPTH, NR = 16 / WK;
    static_assert(NR == 2, "the pair layout of the k-part exchange is for eight k-parts");
    {   // every cache line of the argument block is requested at entry (one scalar round trip instead of three)
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
    const int s0 = sb + (len * wk) / WK, s1 = sb + (len * (wk + 1)) / WK;  // this wave's k64-steps (may be empty)
    const int mrows = a.M;  // 1 .. 32

    const char* wt[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int nt = min(cg * CT + t, a.NT - 1);  // a tile past the matrix re-reads the last one and is never stored
        wt[t] = reinterpret_cast<const char*>(a.prep) + (int64_t)nt * a.KS * 4096;
    }
    const uint32_t woff = lane * 16;
    const int sclamp = max(s1 - 1, s0);  // loads past the wave's steps re-read its last one (a cache hit), never consumed
    const char* xb = reinterpret_cast<const char*>(a.x);

    V8 wq[DEPTH][CT][4];
    V8 xa[DEPTH][4];
    auto load_step = [&](int d, int step) {
        const int sc = min(step, sclamp);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const char* p = wt[t] + (int64_t)sc * 4096;
            PIN_SGPR(p);
#pragma unroll
            for (int i = 0; i < 4; ++i) wq[d][t][i] = __builtin_nontemporal_load((const GLOBAL_AS V8*)(p + woff + i * 1024));
        }
        const char* p = xb + (int64_t)sc * 4096;  // fragment order: four contiguous KiB per k64-step
        PIN_SGPR(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) xa[d][i] = *(const GLOBAL_AS V8*)(p + woff + i * 1024);
    };

    // the rows this wave finishes: accumulator registers [wk NR, (wk + 1) NR) of every tile
    auto row_of = [&](int j) {
        const int r = wk * NR + j;
        return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    };
    // ACT 3: their cache slots and rotary positions
    int32_t rpos[ACT == 3 ? NR : 1], rslot[ACT == 3 ? NR : 1];
    if (ACT == 3) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int m = min(row_of(j), mrows - 1);
            rpos[j] = a.positions[m];
            rslot[j] = a.slots[m];
        }
    }

    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_step(d, s0 + d);

    auto consume = [&](int d) {
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t] = mfma32(xa[d][i], wq[d][t][i], acc[t]);
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
    // the last group: only the steps that exist (wave-uniform branches; nothing is requested any more)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (s + d < s1) consume(d);

    // ACT 3: the finishing rows' cos / sin entries are requested before the exchange (their positions came in at entry)
    T rcos[ACT == 3 ? CT : 1][ACT == 3 ? NR : 1], rsin[ACT == 3 ? CT : 1][ACT == 3 ? NR : 1];
    if (ACT == 3) {
        const int per = a.rD >> 5;
        const T* cosb = reinterpret_cast<const T*>(a.cosb);
        const T* sinb = reinterpret_cast<const T*>(a.sinb);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int nt = min(cg * CT + t, a.NT - 1);
            const int tt = nt - (nt / per) * per;
            const int dr = 16 * tt + (lane & 15);
            const bool roth = nt / per < a.rH + a.rHkv;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                rcos[t][j] = roth ? cosb[(int64_t)rpos[j] * (a.rD >> 1) + dr] : from_f32<T>(1.f);
                rsin[t][j] = roth ? sinb[(int64_t)rpos[j] * (a.rD >> 1) + dr] : from_f32<T>(0.f);
            }
        }
    }

    // ---- k-part sum through LDS: [k-part][tile][register / 2][lane][2] (8-byte accesses, every access 128 consecutive words),
    // then wave wk sums registers [wk NR, (wk + 1) NR) of every tile in the fixed order of the k-parts ----
    float* red = reinterpret_cast<float*>(smem);
    const int c = lane & 31;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        f32x2* dst = reinterpret_cast<f32x2*>(red + ((wk * CT + t) << 10)) + lane;
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) dst[rp << 6] = f32x2{acc[t][2 * rp], acc[t][2 * rp + 1]};
    }
    __syncthreads();
    float fin[CT][NR];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
#pragma unroll
        for (int k2 = 0; k2 < WK; ++k2) {
            const f32x2 v = (reinterpret_cast<const f32x2*>(red + ((k2 * CT + t) << 10)) + (wk << 6))[lane];
#pragma unroll
            for (int j = 0; j < NR; ++j) fin[t][j] = k2 == 0 ? v[j] : fin[t][j] + v[j];
        }
    }
    T* out = reinterpret_cast<T*>(a.out);
    const T* bias = reinterpret_cast<const T*>(a.bias);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int nt = cg * CT + t;
        if (nt >= a.NT) break;
        if (ACT == 3) {
            // rope image: see dense_gemm_body.h (the same epilogue on the same image)
            const int per = a.rD >> 5;
            const int head = nt / per, tt = nt - head * per;
            const bool roth = head < a.rH + a.rHkv;
            const int d = roth ? ((c < 16) ? 16 * tt + c : (a.rD >> 1) + 16 * tt + (c - 16)) : 32 * tt + c;
            const int col = head * a.rD + d;
            const float bv = bias ? to_f32(bias[col]) : 0.f;
            T* kpool = reinterpret_cast<T*>(a.kpool);
            T* vpool = reinterpret_cast<T*>(a.vpool);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int m = row_of(j);
                const float mine = to_f32(from_f32<T>(fin[t][j] + bv));
                float o = mine;
                if (roth) {
                    const float other = __shfl_xor(mine, 16, 64);
                    const float cf = to_f32(rcos[t][j]), sf = to_f32(rsin[t][j]);
                    o = (c < 16) ? mine * cf - other * sf : other * sf + mine * cf;
                }
                const T oh = from_f32<T>(o);
                if (m < mrows) {
                    if (head < a.rH) {
                        out[(int64_t)m * a.ldo + col] = oh;
                    } else {
                        const int page = rslot[j] >> 5, tok = rslot[j] & 31;
                        if (roth)
                            kpool[((int64_t)page * a.rHkv + (head - a.rH)) * 32 * a.rD + k_off(tok, d, a.rD)] = oh;
                        else
                            vpool[((int64_t)page * a.rHkv + (head - a.rH - a.rHkv)) * 32 * a.rD + v_off(tok, d, a.rD)] = oh;
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
            const float bv = (bias && j2 < half) ? to_f32(bias[nsrc]) : 0.f;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const float mine = to_f32(from_f32<T>(fin[t][j] + bv));
                const float other = __shfl_xor(mine, 16, 64);
                const int m = row_of(j);
                if (c < 16 && j2 < half && m < mrows) {
                    const float sl = mine / (1.f + __expf(-mine));
                    const T o = from_f32<T>(to_f32(from_f32<T>(sl)) * other);
                    if (OUTF)
                        out[xf_off(m, j2, half)] = o;
                    else
                        out[(int64_t)m * a.ldo + j2] = o;
                }
            }
            continue;
        }
        if (a.S == 1 && !a.partial) {
            const float bv = (bias && n < a.N) ? to_f32(bias[n]) : 0.f;
            if (n < a.N) {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int m = row_of(j);
                    if (m >= mrows) continue;
                    if (a.out_f32)
                        reinterpret_cast<float*>(a.out)[(int64_t)m * a.ldo + n] = fin[t][j] + bv;
                    else if (OUTF)
                        out[xf_off(m, n, a.N)] = finish_out<T>(fin[t][j] + bv, a.gelu);
                    else
                        out[(int64_t)m * a.ldo + n] = finish_out<T>(fin[t][j] + bv, a.gelu);
                }
            }
        } else {
            // slabs in 32-row units: [row block][split][32][ld]
            float* sl = a.slabs + ((int64_t)split * 32) * (a.NT * 32) + n;
#pragma unroll
            for (int j = 0; j < NR; ++j) sl[(int64_t)row_of(j) * (a.NT * 32)] = fin[t][j];
        }
    }
}

}  // namespace dense
