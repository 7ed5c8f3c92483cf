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
#include "dense_gemm_body.h"

namespace {

// gate_up: W is [gate rows | up rows] (N = 2 I); tile nt then holds gate rows 16 nt .. +15 in columns 0..15 and the
// matching up rows in columns 16..31, so that the ACT = 2 epilogue finds each (gate, up) pair inside one wave.
template <typename T>
__global__ void dense_prepare_kernel(const T* __restrict__ w, T* __restrict__ out, int64_t N, int64_t K,
                                     int64_t NT, int64_t KS, int flags) {
    const int gate_up = flags & 1;
    using V8 = typename VecT<T>::x8;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk each
    if (idx >= NT * KS * 256) return;
    int l = idx & 63;
    int i = (idx >> 6) & 3;
    int64_t ks = (idx >> 8) % KS;
    int64_t nt = (idx >> 8) / KS;
    int64_t n = nt * 32 + (l & 31);
    if (gate_up) {
        const int64_t half = N >> 1, j = nt * 16 + (l & 15);
        n = j < half ? ((l & 31) < 16 ? j : half + j) : N;  // past the last pair: a zero column
    } else if (flags & 2) {
        // rope image of a fused qkv projection (head size D in bits 8..19, rotated heads H + Hkv in bits 20..31): a tile of
        // a rotated head holds dims [16 t, 16 t + 16) in columns 0-15 and their rotation partners D/2 + [16 t, ..) in 16-31
        const int D = (flags >> 8) & 0xFFF, nrot = (flags >> 20) & 0xFFF, per = D >> 5, c = l & 31;
        const int64_t head = nt / per, t = nt - head * per;
        if (head < nrot) n = head * D + ((c < 16) ? 16 * t + c : (D >> 1) + 16 * t + (c - 16));
    }
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

using dense::DenseArgs;
using dense::DensePlan;
using dense::DRS;
using dense::plan_dense;
using dense::dense_slab_bytes;

// One workgroup = one unit of dense_gemm_body.h (block-wide barriers, plain loads and stores).
template <typename T, int TN, int WK, int ACT, int MR, bool R16 = false>
__global__ __launch_bounds__(64 * TN * WK) void dense_gemm_kernel(DenseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    dense::dense_gemm_unit<T, TN, WK, ACT, MR, R16>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}


// Sum the S split-K slabs in fixed order and emit the output (+bias); thread = (row, 4 columns).
template <typename T>
__global__ __launch_bounds__(256) void dense_splitk_reduce_kernel(const float* __restrict__ slabs,
                                                                  const T* __restrict__ bias, void* __restrict__ out,
                                                                  int64_t ldo, int M, int N, int NP, int S, int out_f32,
                                                                  int gelu) {
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
            reinterpret_cast<T*>(out)[o] = dense::finish_out<T>(f, gelu);
    }
}


template <typename T, int TN, int WK, int ACT, int MR, bool R16 = false>
static int launch_dense_one(dim3 grid, size_t lds, hipStream_t st, const DenseArgs& a) {
    static bool attr = false;
    if (!attr) {
        TGIS_CHECK_HIP(hipFuncSetAttribute((const void*)dense_gemm_kernel<T, TN, WK, ACT, MR, R16>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 32 * DRS * 2 + 64));
        attr = true;
    }
    hipLaunchKernelGGL((dense_gemm_kernel<T, TN, WK, ACT, MR, R16>), grid, dim3(64 * TN * WK), lds, st, a);
    return TGIS_OK;
}
template <typename T, int TN, int WK, int ACT>
static int launch_dense_variant(int mr, dim3 grid, size_t lds, hipStream_t st, const DenseArgs& a) {
    if constexpr (WK == 2) {
        if (mr == 2) return launch_dense_one<T, TN, WK, ACT, 2>(grid, lds, st, a);
    }
    // batches of up to 16 rows stage 16 rows of x (dense_gemm_body.h, R16); the LDS request shrinks with them.  Same box,
    // 32-row staging patched back in: cfg2 1.066 / 1.067 -> 1.038 / 1.044 ms per step (profiles/r06d_bench_cfg2_rows16_*.json)
    if (a.M <= 16)
        return launch_dense_one<T, TN, WK, ACT, 1, true>(grid, (size_t)WK * 2 * 16 * DRS * sizeof(T) + 64, st, a);
    return launch_dense_one<T, TN, WK, ACT, 1>(grid, lds, st, a);
}

template <typename T>
static int launch_dense(const DenseArgs& a, const DensePlan& pl, int act, int64_t mslabs32, hipStream_t st) {
    dim3 grid((unsigned)cdiv64(a.NT, pl.TN), (unsigned)pl.S, (unsigned)cdiv64(mslabs32, pl.MR));
    const size_t lds = (size_t)pl.WK * 2 * 32 * pl.MR * DRS * sizeof(T) + 64;
    int rc = TGIS_EINVAL;
#define TGIS_DENSE_CASE(T_, W_)                                                        \
    if (pl.TN == T_ && pl.WK == W_)                                                    \
        rc = act == 3   ? launch_dense_variant<T, T_, W_, 3>(pl.MR, grid, lds, st, a)                                           \
             : act == 2 ? launch_dense_variant<T, T_, W_, 2>(pl.MR, grid, lds, st, a)                                           \
             : act == 1 ? launch_dense_variant<T, T_, W_, 1>(pl.MR, grid, lds, st, a)                                           \
                        : launch_dense_variant<T, T_, W_, 0>(pl.MR, grid, lds, st, a)
    // one tile per k-part group: the fused qkv + rotary launch of a narrow projection (TinyLlama: 80 tiles, unsplit)
    if (pl.TN == 1 && pl.WK == 4 && act == 3) rc = launch_dense_variant<T, 1, 4, 3>(pl.MR, grid, lds, st, a);
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
                           a.M, a.N, NP, a.S, a.out_f32, a.gelu);
        TGIS_CHECK_LAUNCH();
    }
    return TGIS_OK;
}

}  // namespace

extern "C" int64_t tgis_dense_prepared_bytes(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0) return 0;
    return cdiv64(N, 32) * cdiv64(K, 64) * 4096;
}

extern "C" int tgis_dense_prepare(const void* w, int64_t N, int64_t K, int dtype, int flags, void* prepared,
                                  void* stream) {
    TGIS_CHECK_ARG(w && prepared && N > 0 && K > 0, "tgis_dense_prepare: bad arguments");
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_dense_prepare: bad dtype");
    const int gate_up = flags & 1;
    TGIS_CHECK_ARG(!gate_up || N % 32 == 0, "tgis_dense_prepare: the gate|up image needs N / 2 to be a multiple of 16");
    if (flags & 2) {
        const int D = (flags >> 8) & 0xFFF, nrot = (flags >> 20) & 0xFFF;
        TGIS_CHECK_ARG(!gate_up && D >= 32 && D % 32 == 0 && nrot >= 1 && (int64_t)nrot * D <= N && N % D == 0,
                       "tgis_dense_prepare: rope image needs head size %% 32 == 0 and rotated heads within N (D=%d, heads=%d)",
                       D, nrot);
    }
    int64_t NT = cdiv64(N, 32), KS = cdiv64(K, 64);
    int64_t total = NT * KS * 256;
    dim3 grid((unsigned)cdiv64(total, 256));
    if (dtype == TGIS_F16)
        hipLaunchKernelGGL(dense_prepare_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)w,
                           (f16*)prepared, N, K, NT, KS, flags);
    else
        hipLaunchKernelGGL(dense_prepare_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)w,
                           (bf16*)prepared, N, K, NT, KS, flags);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int64_t tgis_dense_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N) {
    DensePlan pl = plan_dense(K, N, M);  // (act 2 never splits: this bound covers it)
    return 4096 + (pl.S > 1 ? dense_slab_bytes(M, N, pl.S) : 0);
}

static int dense_check(const void* x, int64_t ldx, const void* prepared, int64_t M, int64_t K, int64_t N, int dtype,
                       int act) {
    TGIS_CHECK_ARG(x && prepared, "tgis_dense_gemm: null tensor");
    TGIS_CHECK_ARG(M >= 0 && K > 0 && N > 0, "tgis_dense_gemm: bad shape");
    TGIS_CHECK_ARG(K % 8 == 0, "tgis_dense_gemm: K (%ld) must be a multiple of 8", (long)K);
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_dense_gemm: bad dtype");
    TGIS_CHECK_ARG((act >= 0 && act <= 2) || act == 4 || act == 5, "tgis_dense_gemm: act must be 0, 1, 2, 4 or 5");
    TGIS_CHECK_ARG(act != 2 || N % 32 == 0, "tgis_dense_gemm: act 2 needs a gate|up image (N / 2 a multiple of 16)");
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
    a.gelu = 0;
    a.positions = a.slots = nullptr;
    a.cosb = a.sinb = nullptr;
    a.kpool = a.vpool = nullptr;
    a.rH = a.rHkv = a.rD = 0;
}

extern "C" int tgis_dense_gemm(const void* x, int64_t ldx, const void* prepared, const void* bias, void* out,
                               int64_t ldo, int64_t M, int64_t K, int64_t N, int dtype, int out_f32, int act,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    int rc = dense_check(x, ldx, prepared, M, K, N, dtype, act);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(out, "tgis_dense_gemm: null out");
    if (M == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    const int gelu = act == 4 ? 1 : act == 5 ? 2 : 0;  // GELU of the finished sum: epilogue (unsplit) or split-K reduce
    if (gelu) act = 0;
    DensePlan pl = plan_dense(K, N, M, act);
    TGIS_CHECK_ARG(act != 2 || (!out_f32 && pl.S == 1), "tgis_dense_gemm: act 2 writes the model dtype, unsplit");
    TGIS_CHECK_ARG(!gelu || !out_f32, "tgis_dense_gemm: act 4 / 5 (GELU) write the model dtype");
    const int64_t need = tgis_dense_gemm_workspace_bytes(M, K, N);
    TGIS_CHECK_ARG(workspace && workspace_bytes >= need, "tgis_dense_gemm: workspace too small (%ld < %ld)",
                   (long)workspace_bytes, (long)need);
    TgisTimedScope timed(TGIS_OP_DENSE_GEMM, st);
    DenseArgs a;
    dense_fill(a, x, ldx, prepared, bias, out, ldo, M, K, N, out_f32, (float*)((uint8_t*)workspace + 4096), 0, pl);
    a.gelu = gelu;
    return dtype == TGIS_F16 ? launch_dense<f16>(a, pl, act, cdiv64(M, 32), st)
                             : launch_dense<bf16>(a, pl, act, cdiv64(M, 32), st);
}

// ---- qkv projection with the rotary embedding and the cache write in its epilogue (dense weights) ---------------------
// The unsplit plan of the fused qkv + rotary launch: plan_dense's SiLU plan, with ONE tile per k-part group where two would
// leave fewer than 128 blocks (round 5: TinyLlama's 80 tiles ran as 40 blocks of 262 KB each — one CU takes in ~50 GB/s).
static DensePlan plan_dense_rope(int64_t K, int64_t N, int64_t M) {
    DensePlan pl = plan_dense(K, N, M, 2);
    const int64_t tiles = cdiv64(N, 32);
    if (M <= 32 && pl.WK == 4 && cdiv64(tiles, pl.TN) < 128) pl.TN = 1;
    return pl;
}

extern "C" int tgis_dense_rope_ok(int64_t M, int64_t K, int64_t N, int64_t D) {
    if (M < 1 || M > 64 || D < 32 || D % 32 || N <= 0 || N % D || K <= 0 || K % 8) return 0;
    // as tgis_gptq_rope_ok: the unsplit plan must still cover the chip — from 64 blocks on when they are one-tile blocks
    const DensePlan pl = plan_dense_rope(K, N, M);
    const int64_t blocks = cdiv64(cdiv64(N, 32), pl.TN);
    static const int64_t min_blocks = getenv("TGIS_ROPE_MIN_BLOCKS") ? atoll(getenv("TGIS_ROPE_MIN_BLOCKS")) : 128;
    return blocks >= (pl.TN == 1 ? std::min<int64_t>(min_blocks, 64) : min_blocks) ? 1 : 0;
}

extern "C" int tgis_dense_gemm_rope(const void* x, int64_t ldx, const void* prepared, const void* bias,
                                    const int32_t* positions, const int32_t* slots, const void* cos, const void* sin,
                                    void* q_out, int64_t ldq, void* k_pool, void* v_pool, int64_t M, int64_t K, int64_t N,
                                    int64_t H, int64_t Hkv, int64_t D, int dtype, void* stream) {
    int rc = dense_check(x, ldx, prepared, M, K, N, dtype, 0);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(positions && slots && cos && sin && q_out && k_pool && v_pool, "tgis_dense_gemm_rope: null tensor");
    TGIS_CHECK_ARG(M >= 1 && M <= 64 && D >= 32 && D % 32 == 0, "tgis_dense_gemm_rope: needs 1 <= M <= 64 and a head size "
                   "that is a multiple of 32 (M=%ld D=%ld)", (long)M, (long)D);
    TGIS_CHECK_ARG(H >= 1 && Hkv >= 1 && (H + 2 * Hkv) * D == N && ldq >= H * D,
                   "tgis_dense_gemm_rope: N must be (H + 2 Hkv) * D and q rows must hold H * D elements");
    hipStream_t st = (hipStream_t)stream;
    DensePlan pl = plan_dense_rope(K, N, M);  // as the SiLU epilogue: the whole k range in one block (S == 1)
    TgisTimedScope timed(TGIS_OP_DENSE_GEMM, st);
    DenseArgs a;
    dense_fill(a, x, ldx, prepared, bias, q_out, ldq, M, K, N, 0, nullptr, 0, pl);
    a.positions = positions;
    a.slots = slots;
    a.cosb = cos;
    a.sinb = sin;
    a.kpool = k_pool;
    a.vpool = v_pool;
    a.rH = (int)H;
    a.rHkv = (int)Hkv;
    a.rD = (int)D;
    return dtype == TGIS_F16 ? launch_dense<f16>(a, pl, 3, cdiv64(M, 32), st) : launch_dense<bf16>(a, pl, 3, cdiv64(M, 32), st);
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
    TGIS_CHECK_ARG(act == 0 || act == 1, "tgis_dense_gemm_partial: the SiLU * up and GELU epilogues need the finished sum (use tgis_dense_gemm)");
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
