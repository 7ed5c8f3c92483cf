// The dense (f16 / bf16 weights) skinny GEMM as a device function (one workgroup = one unit) plus its launch planning; the
// kernel that runs it is in gemm_dense.hip.  (The persistent decode-tail form of rounds 2 - 5 is gone: experiments/README.md.)
#pragma once
#include "gptq_gemm_body.h"
#include "kv_layout.h"

namespace dense {

using gptq::lds_int;

struct DenseArgs {
    const void* x;
    int64_t ldx;
    const uint8_t* prep;
    const void* bias;
    void* out;
    int64_t ldo;
    int M, K, N;   // M = all rows (grid.z walks 32-row slabs)
    int KR;        // k-range per block (multiple of 256 * WK)
    int S;         // global k splits
    int NT, KS;
    int out_f32;
    float* slabs;  // [Mslabs][S][32][NT*32] f32 partial sums (S > 1 or partial mode)
    int partial;   // 1: always leave fp32 slabs (deferred reduce), never write `out`
    int gelu;      // model-dtype output only: 0 none, 1 GELU (erf), 2 GELU (tanh) applied to the rounded sum + bias
    // ACT == 3 (rope image, tgis_dense_prepare flags bit 1): the epilogue rotates q / k heads and writes k / v into their
    // cache pages; `out` is the q tensor (see tgis_dense_gemm_rope)
    const int32_t* positions;  // [M]
    const int32_t* slots;      // [M] page * 32 + token
    const void* cosb;          // [max_pos][rD / 2], model dtype
    const void* sinb;
    void* kpool;               // [pages][rHkv][32 * rD] in the K page layout of kv_layout.h
    void* vpool;
    int rH, rHkv, rD;
};

constexpr int DKC = 256;      // k per LDS chunk (4 k64-steps)
constexpr int DRS = DKC + 8;  // LDS row stride in elements (+16 B -> conflict-free ds_read_b128)
constexpr int DRING = 2;      // k64-steps of weights in flight per wave (2 x 4 KiB)
static_assert(DKC == gptq::KC && DRS == gptq::RS, "the dense unit shares the LDS layout of the int4 unit");

// sum + bias -> model dtype; with `gelu` the activation of the ROUNDED value (what tgis_gelu would read back), rounded again
template <typename T>
__device__ __forceinline__ T finish_out(float v, int gelu) {
    T t = from_f32<T>(v);
    if (gelu) t = from_f32<T>(gelu_f32(to_f32(t), gelu == 2));
    return t;
}

// Same structure as gptq_gemm_unit without the dequantisation: a unit of TN*WK waves owns 32*TN columns x KR rows; wave
// (tile wn, k-part wk) streams its tile's fragments over its own k-range (4 KiB per k64-step, two steps in flight,
// refilled in place), each k-part group double-buffers 32x256 chunks of x through LDS and paces itself with an LDS
// arrival counter; k-parts are summed through LDS in fixed order; global k-splits leave fp32 slabs for the consumer.
// The image is zero-padded past K and N; x columns past the wave's k-range are zeroed when the chunk is staged (the
// fragments there belong to the next k-part).
// ACT = 1: x is [rows, 2K] (gate | up); the staged operand is silu(gate) * up with the reference's eager rounding.
// ACT = 2: the image was prepared with interleaved gate / up rows (tgis_dense_prepare flags bit 0: columns 0..15 of
//   tile nt are gate rows 16 nt .., columns 16..31 the matching up rows); the epilogue writes
//   out[m][j] = T(T(silu(T gate)) * T up), [rows, N/2] — once per element, where ACT = 1 recomputes the SiLU in every
//   column block of the consumer.  Needs S == 1 (the planner guarantees it).
// MR = 32-row blocks of x per pass (2 for M > 32: every weight fragment then feeds two MFMAs; needs WK = 2).
// R16 (batches of up to 16 rows, MR == 1): only 16 rows of x are staged — half the requests through the CU's address path,
// half the LDS stores and half the LDS footprint per chunk; lanes 16 - 31 of the A fragment re-read rows 0 - 15 (an LDS
// broadcast), so accumulator rows 16 - 31 repeat rows 0 - 15 and are never stored.  Round 6: at TinyLlama's widths the
// activation is a third to a half of what a block takes in, and rows 16 - 31 of a 16-sequence batch were clamped copies.
template <typename T, int TN, int WK, int ACT, int MR, bool R16 = false>
__device__ __forceinline__ void dense_gemm_unit(const DenseArgs& a, const int ntg, const int split, const int mslab,
                                                unsigned char* smem) {
    static_assert(MR == 1 || WK == 2, "64-row passes need the LDS of two k-parts");
    static_assert(WK > 1, "the finish below exchanges k-parts");
    using V8 = typename VecT<T>::x8;
    static_assert(!R16 || MR == 1, "16-row staging is for single 32-row passes");
    constexpr int SLAB = 32 * MR;          // rows of the output this pass owns
    constexpr int XR = R16 ? 16 : SLAB;    // rows of x staged per chunk
    constexpr int GT = 64 * TN;
    constexpr int NJ = (XR * 32 + GT - 1) / GT;
    constexpr int RSTEP = GT / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w % TN, wk = w / TN, ltid = wn * 64 + lane;
    T* xs = reinterpret_cast<T*>(smem) + wk * (2 * XR * DRS);
    const int m0 = mslab * SLAB;
    const int mrows = min(XR, a.M - m0);   // (R16: the host guarantees M <= 16)
    const int krp = a.KR / WK;
    const int k0 = split * a.KR + wk * krp;
    const int k1 = min(a.K, k0 + krp);
    const int nchunks = krp / DKC;
    const int nt_raw = ntg * TN + wn;
    const int nt = min(nt_raw, a.NT - 1);
    const int ks0 = k0 >> 6;
    const int ks_clamp = min(a.KS - 1, max(ks0, ((k1 + 63) >> 6) - 1));

    const char* wtile = reinterpret_cast<const char*>(a.prep) + (int64_t)nt * a.KS * 4096;
    const uint32_t woff = lane * 16;
    V8 wq[DRING][4];  // the weights a wave has in flight
    auto w_load = [&](int step, V8* dst) {
        const char* p = wtile + (int64_t)min(ks0 + step, ks_clamp) * 4096;
        PIN_SGPR(p);  // wave-uniform base in SGPRs: (sgpr base + lane offset) addressing
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = __builtin_nontemporal_load((const GLOBAL_AS V8*)(p + i * 1024 + woff));
    };

    // ACT 3: cache slot and rotary position of the rows this wave will finish (distributed finish below)
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

    // ---- x staging: rows past M read a clamped row (their outputs are never stored); columns past the k-range are
    //      zeroed at store time ---------------------------------------------------------------------------------------
    const T* xbase = reinterpret_cast<const T*>(a.x) + (int64_t)m0 * a.ldx;
    const int srow = ltid >> 5, scol = (ltid & 31) * 8;
    V8 xg[NJ], xu[NJ];
    bool xok;
    uint32_t rowoff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rowoff[j] = (uint32_t)(min(srow + RSTEP * j, mrows - 1) * (int)a.ldx * 2);
    auto stage_load = [&](int chunk) {
        const int kk = k0 + chunk * DKC + scol;
        xok = kk < k1;
        const int kc = min(kk, a.K - 8);
        const char* xb = reinterpret_cast<const char*>(xbase);
        PIN_SGPR(xb);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint32_t off = rowoff[j] + (uint32_t)kc * 2;
            xg[j] = *(const GLOBAL_AS V8*)(xb + off);
            if (ACT == 1) xu[j] = *(const GLOBAL_AS V8*)(xb + (int64_t)a.K * 2 + off);
        }
    };
    auto stage_store = [&](int buf) {
        T* dst = xs + buf * (XR * DRS) + srow * DRS + scol;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            V8 t = xg[j];
            if (ACT == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float g = to_f32(t[e]);
                    float sl = g / (1.f + __expf(-g));
                    // reference rounds silu(gate) to the model dtype before the multiply (eager torch ops)
                    t[e] = from_f32<T>(to_f32(from_f32<T>(sl)) * to_f32(xu[j][e]));
                }
            }
            if (!xok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (T)0.f;
            }
            if (NJ * RSTEP == XR || srow + RSTEP * j < XR) st16(dst + j * RSTEP * DRS, t);
        }
    };

    f32x16 accs[MR][2];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int i = 0; i < 2; ++i) accs[mr][i] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int xoff = (lane & (R16 ? 15 : 31)) * DRS + (lane >> 5) * 32;

    volatile lds_int* sync_cnt = (volatile lds_int*)(smem + (size_t)WK * 2 * XR * DRS * sizeof(T)) + wk;
    // all waves of the unit meet here; the LDS traffic a wave issued before is complete when it arrives
    auto unit_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (wn == 0 && lane == 0) *sync_cnt = 0;
    stage_load(0);  // x first: a wave's loads return in order and this one is L2-resident
    // the barrier that publishes the zeroed counters sits between the x requests and the weight requests of EVERY wave, so
    // that no wave's first x chunk queues behind another wave's HBM requests in the CU's memory pipeline (as in
    // gptq_gemm_unit)
    unit_barrier();
#pragma unroll
    for (int s = 0; s < DRING; ++s) w_load(s, wq[s]);
    auto group_sync = [&](int target) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lds_int*)sync_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__builtin_amdgcn_readfirstlane(*sync_cnt) < target) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    stage_store(0);
    group_sync(TN);

    auto chunk_body = [&](const int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        if (!LAST) stage_load(chunk + 1);
        __builtin_amdgcn_sched_barrier(0);
        const T* xbuf = xs + (chunk & 1) * (XR * DRS) + xoff;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int step = chunk * 4 + s4;
            const T* xk = xbuf + s4 * 64;
            V8* cur = wq[s4 & (DRING - 1)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    V8 av = ld16<V8>(xk + mr * (32 * DRS) + i * 8);
                    accs[mr][i & 1] = mfma32(av, cur[i], accs[mr][i & 1]);
                }
            // the slot is consumed: refill it in place, DRING steps ahead (the last chunk only refills what it
            // will still consume itself)
            if (!LAST || s4 + DRING < 4) {
                __builtin_amdgcn_sched_barrier(0);
                w_load(step + DRING, cur);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (LAST) return;
        stage_store((chunk + 1) & 1);
        group_sync(TN * (chunk + 2));
    };
    for (int chunk = 0; chunk + 1 < nchunks; ++chunk) chunk_body(chunk, std::false_type{});
    chunk_body(nchunks - 1, std::true_type{});
    // ACT 3: the cos / sin entries of the rows this wave will finish, asked for before the exchange
    constexpr int NR = 16 / WK;
    T rcos[ACT == 3 ? MR : 1][ACT == 3 ? NR : 1], rsin[ACT == 3 ? MR : 1][ACT == 3 ? NR : 1];
    if (ACT == 3) {
        const int per = a.rD >> 5;
        const int tt = nt - (nt / per) * per;
        const int dr = 16 * tt + (lane & 15);
        const bool roth = nt / per < a.rH + a.rHkv;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                rcos[mr][j] = roth ? reinterpret_cast<const T*>(a.cosb)[(int64_t)rpos[mr][j] * (a.rD >> 1) + dr] : (T)1.f;
                rsin[mr][j] = roth ? reinterpret_cast<const T*>(a.sinb)[(int64_t)rpos[mr][j] * (a.rD >> 1) + dr] : (T)0.f;
            }
    }
    unit_barrier();  // every k-part is done with its x buffers: the reduction below reuses them

    f32x16 acc[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) acc[mr] = accs[mr][0] + accs[mr][1];
    // ---- distributed finish (as gptq_gemm_unit) ---------------------------------------------------------------------------
    // Every wave leaves its partial sums in LDS; wave (wn, wk) sums the WK k-parts of accumulator registers
    // [wk NR, (wk + 1) NR) of tile wn in the fixed order 0..WK-1 (bit-identical to the reducer-wave form) and runs the
    // epilogue for those rows only.
    {
        float* red = reinterpret_cast<float*>(smem);  // [WK][TN tiles][MR][64 lanes][16]
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            float* dst = red + ((((wk * TN + wn) * MR + mr) * 64 + lane) << 4);
#pragma unroll
            for (int r = 0; r < 16; r += 4)
                *reinterpret_cast<f32x4*>(dst + r) = f32x4{acc[mr][r], acc[mr][r + 1], acc[mr][r + 2], acc[mr][r + 3]};
        }
        unit_barrier();
        if (nt_raw >= a.NT) return;
        float fin[MR][NR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
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
        const int c = lane & 31;
        auto row_of = [&](int j) {
            const int r = wk * NR + j;
            return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        };
        if (ACT == 3) {
            // rope image: a tile of a q / k head holds dims [16 t, 16 t + 16) in lanes c < 16 and their rotation partners
            // rD/2 + [16 t, ..) in lanes c + 16; v heads keep 32 consecutive dims.  Sum (+ bias) rounded to T, rotated in
            // fp32 (the arithmetic of rope_kv_kernel), q to `out`, k / v into their cache pages.  Host guarantees S == 1.
            const int per = a.rD >> 5;
            const int head = nt / per, tt = nt - head * per;
            const bool roth = head < a.rH + a.rHkv;
            const int d = roth ? ((c < 16) ? 16 * tt + c : (a.rD >> 1) + 16 * tt + (c - 16)) : 32 * tt + c;
            const int col = head * a.rD + d;
            const float bv = a.bias ? to_f32(reinterpret_cast<const T*>(a.bias)[col]) : 0.f;
            T* kpool = reinterpret_cast<T*>(a.kpool);
            T* vpool = reinterpret_cast<T*>(a.vpool);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int m = mr * 32 + row_of(j);
                    const float mine = to_f32(from_f32<T>(fin[mr][j] + bv));
                    float o = mine;
                    if (roth) {
                        const float other = __shfl_xor(mine, 16, 64);
                        const float cf = to_f32(rcos[mr][j]), sf = to_f32(rsin[mr][j]);
                        o = (c < 16) ? mine * cf - other * sf : other * sf + mine * cf;
                    }
                    const T oh = from_f32<T>(o);
                    if (m < mrows) {
                        if (head < a.rH) {
                            reinterpret_cast<T*>(a.out)[(int64_t)(m0 + m) * a.ldo + col] = oh;
                        } else {
                            const int page = rslot[mr][j] >> 5, tok = rslot[mr][j] & 31;
                            if (roth)
                                kpool[((int64_t)page * a.rHkv + (head - a.rH)) * 32 * a.rD + k_off(tok, d, a.rD)] = oh;
                            else
                                vpool[((int64_t)page * a.rHkv + (head - a.rH - a.rHkv)) * 32 * a.rD + v_off(tok, d, a.rD)] = oh;
                        }
                    }
                }
            return;
        }
        const int n = nt * 32 + c;
        if (ACT == 2) {
            const int half = a.N >> 1;
            const int j2 = nt * 16 + (c & 15);
            const int nsrc = (c < 16) ? j2 : half + j2;
            const float bv = (a.bias && j2 < half) ? to_f32(reinterpret_cast<const T*>(a.bias)[nsrc]) : 0.f;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const float mine = to_f32(from_f32<T>(fin[mr][j] + bv));
                    const float other = __shfl_xor(mine, 16, 64);
                    const int m = mr * 32 + row_of(j);
                    if (c < 16 && j2 < half && m < mrows) {
                        float sl = mine / (1.f + __expf(-mine));
                        reinterpret_cast<T*>(a.out)[(int64_t)(m0 + m) * a.ldo + j2] =
                            from_f32<T>(to_f32(from_f32<T>(sl)) * other);
                    }
                }
            return;
        }
        if (a.S == 1 && !a.partial) {
            if (n >= a.N) return;
            const float bv = a.bias ? to_f32(reinterpret_cast<const T*>(a.bias)[n]) : 0.f;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int m = mr * 32 + row_of(j);
                    if (m >= mrows) continue;
                    if (a.out_f32)
                        reinterpret_cast<float*>(a.out)[(int64_t)(m0 + m) * a.ldo + n] = fin[mr][j] + bv;
                    else
                        reinterpret_cast<T*>(a.out)[(int64_t)(m0 + m) * a.ldo + n] = finish_out<T>(fin[mr][j] + bv, a.gelu);
                }
        } else {
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

struct DensePlan {
    int KR, S, WK, TN, MR;
};

// Plan with at most `max_units` units (one per CU).  `direct`: the output is finished in the epilogue (SiLU * up, rotary +
// cache write), so no global k-split — the k-range is spread over four in-block k-parts instead.
static inline DensePlan plan_dense_tail(int64_t K, int64_t N, bool direct, int64_t max_units) {
    const int64_t tiles = cdiv64(N, 32);
    const int64_t kchunks = cdiv64(K, DKC);
    int TN = 2;
    while (TN < 4 && cdiv64(tiles, TN) > max_units) ++TN;
    const int64_t colblocks = cdiv64(tiles, TN);
    int WK;
    int64_t S = 1;
    if (direct) {
        WK = kchunks >= 4 ? 4 : 2;
    } else {
        S = std::max<int64_t>(1, std::min<int64_t>(kchunks, max_units / std::max<int64_t>(colblocks, 1)));
        while (S > 1 && (S - 1) * cdiv64(kchunks, S) >= kchunks) --S;
        WK = cdiv64(kchunks, S) >= 4 ? 4 : 2;
    }
    int64_t KRc = cdiv64(kchunks, S);
    if (KRc < WK) WK = 2;
    KRc = cdiv64(KRc, WK) * WK;
    while (S > 1 && (S - 1) * KRc >= kchunks) --S;
    return {(int)(KRc * DKC), (int)S, WK, TN, 1};
}

// Same shape rules as plan_gemm (gptq_gemm_body.h) — the bytes per tile are 4x, the block structure is the same.
// act == 2 (SiLU * up epilogue) needs the whole k-sum in one unit: never a global k-split.
static inline DensePlan plan_dense(int64_t K, int64_t N, int64_t M = 32, int act = 0) {
    const int64_t tiles = cdiv64(N, 32);
    const int64_t kchunks = cdiv64(K, DKC);
    const int MR = M > 32 ? 2 : 1;
    int TN, WK;
    int64_t S = 1;
    if (MR == 2) {
        TN = tiles >= 256 ? 4 : 2;
        WK = 2;
        const int64_t colblocks = cdiv64(tiles, TN) * cdiv64(M, 64);
        S = std::max<int64_t>(1, std::min<int64_t>(kchunks / 2, (224 + colblocks / 2) / colblocks));
        while (S > 1 && (S - 1) * cdiv64(kchunks, S) >= kchunks) --S;
        if (act == 2) S = 1;
    } else if (tiles >= 512) {
        TN = cdiv64(tiles, 3) <= 256 ? 3 : 4;
        WK = 4;
    } else if (act == 2) {
        return plan_dense_tail(K, N, true, 256);
    } else if (K * N * 2 < (48ll << 20)) {
        // small matrices are latency-bound: as many blocks as one round holds, short k-parts (TinyLlama sweeps)
        TN = tiles >= 256 ? 4 : 2;
        const int64_t colblocks = cdiv64(tiles, TN);
        const int64_t want = TN == 4 ? 224 : 256;
        S = std::max<int64_t>(1, std::min<int64_t>(kchunks, (want + colblocks / 2) / colblocks));
        while (S > 1 && (S - 1) * cdiv64(kchunks, S) >= kchunks) --S;  // no empty last split
        WK = TN == 4 ? 2 : (cdiv64(kchunks, S) >= 4 ? 4 : 2);
    } else {
        // Narrow / medium N, bandwidth-bound sizes: pick (TN, WK, S) by a two-term model of a launch — a dense block streams
        // KR x 32 TN x 2 bytes; blocks run in rounds of one per CU (two for the half-size LDS of WK = 2); a round takes
        // max(block bytes / per-CU rate, round bytes / chip rate) + a fixed ramp; each extra split adds slab traffic.
        // The rates are the ones measured on MI355X (one block alone ~50 GB/s, the chip 5.9 TB/s for this access
        // pattern).  It reproduces the measured 72 us of 24576x6144 at (2,4,S=3: 288 blocks = two rounds) and picks
        // (3,4,S=4: 256 blocks, one round) instead.
        double best = 1e30;
        TN = 2, WK = 4, S = 1;
        for (int tn = 2; tn <= 4; ++tn)
            for (int wk = 4; wk >= 2; wk -= 2) {
                if (tn == 3 && wk == 2) continue;  // not instantiated
                for (int64_t sp = 1; sp <= std::min<int64_t>(16, kchunks); ++sp) {
                    int64_t krc = cdiv64(cdiv64(kchunks, sp), wk) * wk;
                    if (sp > 1 && (sp - 1) * krc >= kchunks) continue;  // an empty last split
                    const int64_t blocks = cdiv64(tiles, tn) * sp;
                    const double block_bytes = (double)krc * DKC * tn * 32 * 2;
                    const int64_t slots = wk == 2 ? 512 : 256;
                    const double cu_rate = wk == 2 ? 25.0 : 50.0;  // GB/s per block: two half-LDS blocks share a CU
                    double ns = sp > 1 ? 500.0 * sp : 0.0;
                    for (int64_t left = blocks; left > 0; left -= slots) {
                        const int64_t n = std::min(left, slots);
                        ns += std::max(block_bytes / cu_rate, n * block_bytes / 5900.0) + 4000.0;
                    }
                    if (ns < best - 1.0) {
                        best = ns;
                        TN = tn, WK = wk, S = sp;
                    }
                }
            }
        if (const char* ov = getenv("TGIS_DENSE_PLAN")) {  // tuning hook: "S,WK,TN"
            int sp = 0, wk = 0, tn = 0;
            if (sscanf(ov, "%d,%d,%d", &sp, &wk, &tn) == 3 && sp >= 1 && (wk == 2 || wk == 4) && tn >= 2 && tn <= 4 &&
                !(tn == 3 && wk == 2))
                TN = tn, WK = wk, S = sp;
        }
    }
    int64_t KRc = cdiv64(kchunks, S);
    if (KRc < WK) WK = 2;
    if (TN == 3 && WK == 2) TN = 4;  // short K under a wide N: (3, 2) is not instantiated
    KRc = cdiv64(KRc, WK) * WK;
    while (S > 1 && (S - 1) * KRc >= kchunks) --S;
    return {(int)(KRc * DKC), (int)S, WK, TN, MR};
}

// slabs are stored in 32-row units; a 64-row pass always writes both of its units
static inline int64_t dense_slab_bytes(int64_t M, int64_t N, int S) {
    return cdiv64(M, 64) * 2 * S * 32 * cdiv64(N, 32) * 32 * 4;
}

}  // namespace dense
