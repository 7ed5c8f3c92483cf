// Dense skinny GEMM for decode-sized M (f16 / bf16 weights), gfx950 MFMA 32x32x16.
// Replaces the cuBLAS calls the reference makes through F.linear / torch.mm at decode time:
// FastLinear (utils/layers.py:110-111) and the lm_head matmul (utils/layers.py:261).
//
// Prepared image: W[N,K] (torch Linear layout) repacked into 32-column MFMA tiles so that every
// wave load is one contiguous KiB:  [NT=ceil(N/32)][KS=ceil(K/64)][4][64 lanes][8 elems];
// lane l, word i = W[n = nt*32 + (l&31)][k = (ks*8 + (l>>5)*4 + i)*8 .. +7].
#include <algorithm>
#include <type_traits>
#include "common.h"

namespace {

template <typename T>
__global__ void dense_prepare_kernel(const T* __restrict__ w, T* __restrict__ out, int64_t N, int64_t K,
                                     int64_t NT, int64_t KS) {
    using V8 = typename VecT<T>::x8;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk each
    if (idx >= NT * KS * 256) return;
    int l = idx & 63;
    int i = (idx >> 6) & 3;
    int64_t ks = (idx >> 8) % KS;
    int64_t nt = (idx >> 8) / KS;
    int64_t n = nt * 32 + (l & 31);
    int64_t k = (ks * 8 + (l >> 5) * 4 + i) * 8;
    V8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (T)0.f;
    if (n < N) {
        if (k + 8 <= K && (K % 8) == 0) {
            v = ld16<V8>(w + n * K + k);
        } else {
            for (int e = 0; e < 8; ++e)
                if (k + e < K) v[e] = w[n * K + k + e];
        }
    }
    st16(out + idx * 8, v);
}

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
};

constexpr int DKC = 256;      // k per LDS chunk (4 k64-steps)
constexpr int DRS = DKC + 8;  // LDS row stride in elements (+16 B -> conflict-free ds_read_b128)
constexpr int DRING = 2;      // k64-steps of weights in flight per wave (2 x 4 KiB)
#define DENSE_GLOBAL_AS __attribute__((address_space(1)))

// Same structure as gptq_gemm_kernel (gptq.hip) without the dequantisation: a block of TN*WK waves owns 32*TN columns
// x KR rows; wave (tile wn, k-part wk) streams its tile's fragments over its own k-range (4 KiB per k64-step, two
// steps in flight, refilled in place), each k-part group double-buffers 32x256 chunks of x through LDS and paces
// itself with an LDS arrival counter; k-parts are summed through LDS in fixed order; global k-splits leave fp32
// slabs for the consumer kernel.  The image is zero-padded past K and N; x columns past the wave's k-range are
// zeroed when the chunk is staged (the fragments there belong to the next k-part).
// MR = 32-row blocks of x per pass (2 for M > 32: every weight fragment then feeds two MFMAs instead of being streamed
// again for rows 32..63; needs WK = 2 for the doubled x buffers).
template <typename T, int TN, int WK, int ACT, int MR>
__global__ __launch_bounds__(64 * TN * WK) void dense_gemm_kernel(DenseArgs a) {
    static_assert(MR == 1 || WK == 2, "64-row passes need the LDS of two k-parts");
    using V8 = typename VecT<T>::x8;
    constexpr int XR = 32 * MR;
    constexpr int GT = 64 * TN;
    constexpr int NJ = (XR * 32 + GT - 1) / GT;
    constexpr int RSTEP = GT / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w % TN, wk = w / TN, ltid = wn * 64 + lane;
    T* xs = reinterpret_cast<T*>(smem) + wk * (2 * XR * DRS);
    const int ntg = blockIdx.x, split = blockIdx.y, mslab = blockIdx.z;
    const int m0 = mslab * XR;
    const int mrows = min(XR, a.M - m0);
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
    V8 wq[DRING][4];
    auto w_load = [&](int step, V8* dst) {
        const char* p = wtile + (int64_t)min(ks0 + step, ks_clamp) * 4096;
        asm volatile("" : "+s"(p));  // wave-uniform base in SGPRs: (sgpr base + lane offset) addressing
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dst[i] = __builtin_nontemporal_load((const DENSE_GLOBAL_AS V8*)(p + i * 1024 + woff));
    };

    // ---- x staging (as in gptq.hip): rows past M read a clamped row (their outputs are never stored); columns past
    //      the k-range are zeroed at store time --------------------------------------------------------------------
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
        asm volatile("" : "+s"(xb));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint32_t off = rowoff[j] + (uint32_t)kc * 2;
            xg[j] = *(const DENSE_GLOBAL_AS V8*)(xb + off);
            if (ACT == 1) xu[j] = *(const DENSE_GLOBAL_AS V8*)(xb + (int64_t)a.K * 2 + off);
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
    const int xoff = (lane & 31) * DRS + (lane >> 5) * 32;

    typedef __attribute__((address_space(3))) int lds_int;
    volatile lds_int* sync_cnt = (volatile lds_int*)(smem + (size_t)WK * 2 * XR * DRS * sizeof(T)) + wk;
    if (wn == 0 && lane == 0) *sync_cnt = 0;
    stage_load(0);  // x first: a wave's loads return in order and this one is L2-resident
#pragma unroll
    for (int s = 0; s < DRING; ++s) w_load(s, wq[s]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // publishes the zeroed counters; does not wait for the loads above
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every k-part is done with its x buffers: the reduction below reuses them

    f32x16 acc[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) acc[mr] = accs[mr][0] + accs[mr][1];
    if (WK > 1) {
        float* red = reinterpret_cast<float*>(smem);  // [WK][TN tiles][MR][64 lanes][16]
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
    if (nt_raw >= a.NT) return;
    const int n = nt * 32 + (lane & 31);
    if (a.S == 1 && !a.partial) {
        if (n >= a.N) return;
        const float bv = a.bias ? to_f32(reinterpret_cast<const T*>(a.bias)[n]) : 0.f;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= mrows) continue;
                if (a.out_f32)
                    reinterpret_cast<float*>(a.out)[(int64_t)(m0 + m) * a.ldo + n] = acc[mr][r] + bv;
                else
                    reinterpret_cast<T*>(a.out)[(int64_t)(m0 + m) * a.ldo + n] = from_f32<T>(acc[mr][r] + bv);
            }
    } else {
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

// Sum the S split-K slabs in fixed order and emit the output (+bias); thread = (row, 4 columns).
template <typename T>
__global__ __launch_bounds__(256) void dense_splitk_reduce_kernel(const float* __restrict__ slabs,
                                                                  const T* __restrict__ bias, void* __restrict__ out,
                                                                  int64_t ldo, int M, int N, int NP, int S, int out_f32) {
    const int np4 = NP >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int mslab = blockIdx.y;
    if (idx >= (int64_t)32 * np4) return;
    const int m = idx / np4, c4 = (idx - (int64_t)m * np4) * 4;
    if (mslab * 32 + m >= M) return;
    f32x4 v = {0, 0, 0, 0};
    const float* base = slabs + ((int64_t)mslab * S * 32 + m) * NP + c4;
    for (int s2 = 0; s2 < S; ++s2) v += *reinterpret_cast<const f32x4*>(base + (int64_t)s2 * 32 * NP);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (c4 + e >= N) continue;
        float f = v[e] + (bias ? to_f32(bias[c4 + e]) : 0.f);
        const int64_t o = (int64_t)(mslab * 32 + m) * ldo + c4 + e;
        if (out_f32)
            reinterpret_cast<float*>(out)[o] = f;
        else
            reinterpret_cast<T*>(out)[o] = from_f32<T>(f);
    }
}

struct DensePlan {
    int KR, S, WK, TN, MR;
};

// Same shape rules as plan_gemm (gptq.hip) — the bytes per tile are 4x, the block structure is the same.
static DensePlan plan_dense(int64_t K, int64_t N, int64_t M = 32) {
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
    } else if (tiles >= 512) {
        TN = cdiv64(tiles, 3) <= 256 ? 3 : 4;
        WK = 4;
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
    KRc = cdiv64(KRc, WK) * WK;
    while (S > 1 && (S - 1) * KRc >= kchunks) --S;
    return {(int)(KRc * DKC), (int)S, WK, TN, MR};
}

// slabs are stored in 32-row units; a 64-row pass always writes both of its units
static int64_t dense_slab_bytes(int64_t M, int64_t N, int S) {
    return cdiv64(M, 64) * 2 * S * 32 * cdiv64(N, 32) * 32 * 4;
}

template <typename T, int TN, int WK, int ACT, int MR>
static int launch_dense_one(dim3 grid, size_t lds, hipStream_t st, const DenseArgs& a) {
    static bool attr = false;
    if (!attr) {
        TGIS_CHECK_HIP(hipFuncSetAttribute((const void*)dense_gemm_kernel<T, TN, WK, ACT, MR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 32 * DRS * 2 + 64));
        attr = true;
    }
    hipLaunchKernelGGL((dense_gemm_kernel<T, TN, WK, ACT, MR>), grid, dim3(64 * TN * WK), lds, st, a);
    return TGIS_OK;
}
template <typename T, int TN, int WK, int ACT>
static int launch_dense_variant(int mr, dim3 grid, size_t lds, hipStream_t st, const DenseArgs& a) {
    if constexpr (WK == 2) {
        if (mr == 2) return launch_dense_one<T, TN, WK, ACT, 2>(grid, lds, st, a);
    }
    return launch_dense_one<T, TN, WK, ACT, 1>(grid, lds, st, a);
}

template <typename T>
static int launch_dense(const DenseArgs& a, const DensePlan& pl, int act, int64_t mslabs32, hipStream_t st) {
    dim3 grid((unsigned)cdiv64(a.NT, pl.TN), (unsigned)pl.S, (unsigned)cdiv64(mslabs32, pl.MR));
    const size_t lds = (size_t)pl.WK * 2 * 32 * pl.MR * DRS * sizeof(T) + 64;
    int rc = TGIS_EINVAL;
#define TGIS_DENSE_CASE(T_, W_)                                                        \
    if (pl.TN == T_ && pl.WK == W_)                                                    \
        rc = act ? launch_dense_variant<T, T_, W_, 1>(pl.MR, grid, lds, st, a) : launch_dense_variant<T, T_, W_, 0>(pl.MR, grid, lds, st, a)
    TGIS_DENSE_CASE(2, 2);
    TGIS_DENSE_CASE(2, 4);
    TGIS_DENSE_CASE(3, 4);
    TGIS_DENSE_CASE(4, 2);
    TGIS_DENSE_CASE(4, 4);
#undef TGIS_DENSE_CASE
    if (rc != TGIS_OK) {
        tgis_set_error("tgis_dense_gemm: no kernel for plan TN=%d WK=%d", pl.TN, pl.WK);
        return rc;
    }
    TGIS_CHECK_LAUNCH();
    if (!a.partial && pl.S > 1) {
        const int NP = a.NT * 32;
        dim3 rgrid((unsigned)cdiv64((int64_t)32 * (NP / 4), 256), (unsigned)mslabs32);
        hipLaunchKernelGGL(dense_splitk_reduce_kernel<T>, rgrid, dim3(256), 0, st, a.slabs, (const T*)a.bias, a.out, a.ldo,
                           a.M, a.N, NP, a.S, a.out_f32);
        TGIS_CHECK_LAUNCH();
    }
    return TGIS_OK;
}

}  // namespace

extern "C" int64_t tgis_dense_prepared_bytes(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0) return 0;
    return cdiv64(N, 32) * cdiv64(K, 64) * 4096;
}

extern "C" int tgis_dense_prepare(const void* w, int64_t N, int64_t K, int dtype, void* prepared, void* stream) {
    TGIS_CHECK_ARG(w && prepared && N > 0 && K > 0, "tgis_dense_prepare: bad arguments");
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_dense_prepare: bad dtype");
    int64_t NT = cdiv64(N, 32), KS = cdiv64(K, 64);
    int64_t total = NT * KS * 256;
    dim3 grid((unsigned)cdiv64(total, 256));
    if (dtype == TGIS_F16)
        hipLaunchKernelGGL(dense_prepare_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)w,
                           (f16*)prepared, N, K, NT, KS);
    else
        hipLaunchKernelGGL(dense_prepare_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)w,
                           (bf16*)prepared, N, K, NT, KS);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int64_t tgis_dense_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N) {
    DensePlan pl = plan_dense(K, N, M);
    return 4096 + (pl.S > 1 ? dense_slab_bytes(M, N, pl.S) : 0);
}

static int dense_check(const void* x, int64_t ldx, const void* prepared, int64_t M, int64_t K, int64_t N, int dtype,
                       int act) {
    TGIS_CHECK_ARG(x && prepared, "tgis_dense_gemm: null tensor");
    TGIS_CHECK_ARG(M >= 0 && K > 0 && N > 0, "tgis_dense_gemm: bad shape");
    TGIS_CHECK_ARG(K % 8 == 0, "tgis_dense_gemm: K (%ld) must be a multiple of 8", (long)K);
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_dense_gemm: bad dtype");
    TGIS_CHECK_ARG(act == 0 || act == 1, "tgis_dense_gemm: act must be 0 or 1");
    TGIS_CHECK_ARG(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0, "tgis_dense_gemm: x rows must be 16-byte aligned");
    return TGIS_OK;
}

static void dense_fill(DenseArgs& a, const void* x, int64_t ldx, const void* prepared, const void* bias, void* out,
                       int64_t ldo, int64_t M, int64_t K, int64_t N, int out_f32, float* slabs, int partial,
                       const DensePlan& pl) {
    a.x = x;
    a.ldx = ldx;
    a.prep = (const uint8_t*)prepared;
    a.bias = bias;
    a.out = out;
    a.ldo = ldo;
    a.M = (int)M;
    a.K = (int)K;
    a.N = (int)N;
    a.KR = pl.KR;
    a.S = pl.S;
    a.NT = (int)cdiv64(N, 32);
    a.KS = (int)cdiv64(K, 64);
    a.out_f32 = out_f32;
    a.slabs = slabs;
    a.partial = partial;
}

extern "C" int tgis_dense_gemm(const void* x, int64_t ldx, const void* prepared, const void* bias, void* out,
                               int64_t ldo, int64_t M, int64_t K, int64_t N, int dtype, int out_f32, int act,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    int rc = dense_check(x, ldx, prepared, M, K, N, dtype, act);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(out, "tgis_dense_gemm: null out");
    if (M == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    DensePlan pl = plan_dense(K, N, M);
    const int64_t need = tgis_dense_gemm_workspace_bytes(M, K, N);
    TGIS_CHECK_ARG(workspace && workspace_bytes >= need, "tgis_dense_gemm: workspace too small (%ld < %ld)",
                   (long)workspace_bytes, (long)need);
    TgisTimedScope timed(TGIS_OP_DENSE_GEMM, st);
    DenseArgs a;
    dense_fill(a, x, ldx, prepared, bias, out, ldo, M, K, N, out_f32, (float*)((uint8_t*)workspace + 4096), 0, pl);
    return dtype == TGIS_F16 ? launch_dense<f16>(a, pl, act, cdiv64(M, 32), st)
                             : launch_dense<bf16>(a, pl, act, cdiv64(M, 32), st);
}

extern "C" int64_t tgis_dense_gemm_partial_bytes(int64_t M, int64_t K, int64_t N) {
    DensePlan pl = plan_dense(K, N, M);
    return dense_slab_bytes(std::max<int64_t>(M, 1), N, pl.S);
}

extern "C" int tgis_dense_gemm_partial(const void* x, int64_t ldx, const void* prepared, int64_t M, int64_t K,
                                       int64_t N, int dtype, int act, float* slabs, int64_t slabs_bytes,
                                       int* num_slabs, int64_t* slab_ld, void* stream) {
    int rc = dense_check(x, ldx, prepared, M, K, N, dtype, act);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(M >= 1 && cdiv64(M, 32) <= 65535, "tgis_dense_gemm_partial: bad M");
    TGIS_CHECK_ARG(slabs && slabs_bytes >= tgis_dense_gemm_partial_bytes(M, K, N),
                   "tgis_dense_gemm_partial: slab buffer too small");
    hipStream_t st = (hipStream_t)stream;
    DensePlan pl = plan_dense(K, N, M);
    if (num_slabs) *num_slabs = pl.S;
    if (slab_ld) *slab_ld = cdiv64(N, 32) * 32;
    TgisTimedScope timed(TGIS_OP_DENSE_GEMM, st);
    DenseArgs a;
    dense_fill(a, x, ldx, prepared, nullptr, nullptr, 0, M, K, N, 0, slabs, 1, pl);
    return dtype == TGIS_F16 ? launch_dense<f16>(a, pl, act, cdiv64(M, 32), st)
                             : launch_dense<bf16>(a, pl, act, cdiv64(M, 32), st);
}
