// Host side of the lean int4 GEMM (gptq_lean_body.h) and the stand-alone producer of the row sums it consumes.
//
// Replaces exllamav2_kernels.gemm_half_q_half (utils/gptq/exllamav2.py:124-144) for decode batches of up to 32 rows on
// group-size-128 images without act-order; everything else keeps gptq_gemm_kernel (gptq.hip).
#include <stdlib.h>
#include "common.h"
#include "../include/tgis_experiments.h"
#include "gptq_ld_body.h"

namespace {

using gptq::GemmArgs;
using gptq::GemmPlan;
using gptq::LeanArgs;
using gptq::PrepLayout;

template <int TN, int WK, int ACT, int RING>
__global__ __launch_bounds__(64 * TN * WK) void gptq_lean_kernel(LeanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gptq::gptq_lean_unit<TN, WK, ACT, RING>(a, blockIdx.x, blockIdx.y, smem);
}

__device__ unsigned g_ld_err;  // code left by a bounded spin of the loader / consumer kernel that gave up (0: never)

template <int TN, int WK, int ACT, int D>
__global__ __launch_bounds__(64 * (TN * WK + 1)) void gptq_ld_kernel(LeanArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    a.g.err = &g_ld_err;
    gptq::gptq_ld_unit<TN, WK, ACT, D>(a, blockIdx.x, blockIdx.y, smem);
}

// {XA, XB} per row and 16 columns of an f16 matrix: XA = sum of x[k] over k % 4 < 2, XB over k % 4 >= 2 (fp32, fixed
// order: four elements per thread, then the pair of threads that share a 16-column block).
__global__ __launch_bounds__(256) void xsum_kernel(const f16* __restrict__ x, int64_t ldx, float* __restrict__ xs,
                                                   int64_t ldxs, int K) {
    const int64_t row = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;  // 8-column piece
    const bool on = c * 8 < K;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (on) v = ld16<f16x8>(x + row * ldx + (int64_t)c * 8);
    float xa = ((float)v[0] + (float)v[1]) + ((float)v[4] + (float)v[5]);
    float xb = ((float)v[2] + (float)v[3]) + ((float)v[6] + (float)v[7]);
    xa += gptq::dpp_quad_swap1(xa);
    xb += gptq::dpp_quad_swap1(xb);
    if (on && (threadIdx.x & 1) == 0) *reinterpret_cast<f32x2*>(xs + (row * ldxs + (c >> 1)) * 2) = f32x2{xa, xb};
}

int ring_depth() {
    static const int v = getenv("TGIS_LEAN_RING") ? atoi(getenv("TGIS_LEAN_RING")) : 4;
    return v == 8 ? 8 : 4;
}

template <int TN, int WK, int ACT, int RING>
int launch_one(dim3 grid, hipStream_t st, const LeanArgs& a) {
    static bool attr_done = false;
    if (!attr_done) {
        TGIS_CHECK_HIP(hipFuncSetAttribute((const void*)gptq_lean_kernel<TN, WK, ACT, RING>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, gptq::lean_lds_bytes(4)));
        attr_done = true;
    }
    hipLaunchKernelGGL((gptq_lean_kernel<TN, WK, ACT, RING>), grid, dim3(64 * TN * WK), gptq::lean_lds_bytes(WK), st, a);
    return TGIS_OK;
}

// 0: register-ring kernel, 1: loader / consumer kernel (LDS-DMA ring)
int use_ld() {
    static const int v = getenv("TGIS_LEAN_LD") ? atoi(getenv("TGIS_LEAN_LD")) : 0;
    return v;
}
int ld_depth() {
    static const int v = getenv("TGIS_LEAN_D") ? atoi(getenv("TGIS_LEAN_D")) : 4;
    return v == 8 ? 8 : 4;
}

template <int TN, int WK, int ACT, int D>
int launch_ld_one(dim3 grid, hipStream_t st, const LeanArgs& a) {
    constexpr int lds = gptq::ld_lds_bytes(TN * WK, WK, D);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        TGIS_CHECK_HIP(hipFuncSetAttribute((const void*)gptq_ld_kernel<TN, WK, ACT, D>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((gptq_ld_kernel<TN, WK, ACT, D>), grid, dim3(64 * (TN * WK + 1)), lds, st, a);
    return TGIS_OK;
}

template <int ACT>
int launch_ld(const GemmPlan& pl, dim3 grid, hipStream_t st, const LeanArgs& a) {
    const bool deep = ld_depth() == 8;
    switch (pl.TN * 10 + pl.WK) {
        case 34: return launch_ld_one<3, 4, ACT, 4>(grid, st, a);
        case 42: return deep ? launch_ld_one<4, 2, ACT, 8>(grid, st, a) : launch_ld_one<4, 2, ACT, 4>(grid, st, a);
        case 32: return deep ? launch_ld_one<3, 2, ACT, 8>(grid, st, a) : launch_ld_one<3, 2, ACT, 4>(grid, st, a);
        case 24: return deep ? launch_ld_one<2, 4, ACT, 8>(grid, st, a) : launch_ld_one<2, 4, ACT, 4>(grid, st, a);
        default: return deep ? launch_ld_one<2, 2, ACT, 8>(grid, st, a) : launch_ld_one<2, 2, ACT, 4>(grid, st, a);
    }
}

template <int ACT, int RING>
int launch_tw(const GemmPlan& pl, dim3 grid, hipStream_t st, const LeanArgs& a) {
    switch (pl.TN * 10 + pl.WK) {
        case 44: return launch_one<4, 4, ACT, RING>(grid, st, a);
        case 42: return launch_one<4, 2, ACT, RING>(grid, st, a);
        case 34: return launch_one<3, 4, ACT, RING>(grid, st, a);
        case 32: return launch_one<3, 2, ACT, RING>(grid, st, a);
        case 24: return launch_one<2, 4, ACT, RING>(grid, st, a);
        default: return launch_one<2, 2, ACT, RING>(grid, st, a);
    }
}

int launch_lean(const void* x, int64_t ldx, const float* xs, int64_t ldxs, const void* prepared, const void* bias,
                void* out, int64_t ldo, float* xs_out, int64_t M, int64_t K, int64_t N, int64_t groups, int act,
                float* slabs, int partial, const GemmPlan& pl, hipStream_t st) {
    PrepLayout p = gptq::prep_layout(K, N, groups);
    LeanArgs la;
    GemmArgs& a = la.g;
    a.x = (const f16*)x;
    a.ldx = ldx;
    a.prep = (const uint8_t*)prepared;
    a.offB = p.offB;
    a.bias = (const f16*)bias;
    a.perm = nullptr;
    a.out = (f16*)out;
    a.ldo = ldo;
    a.M = (int)M;
    a.K = (int)K;
    a.N = (int)N;
    a.G = (int)groups;
    a.gs = 128;
    a.KR = pl.KR;
    a.S = pl.S;
    a.NT = (int)p.NT;
    a.KS = (int)p.KS;
    a.slabs = slabs;
    a.partial = partial;
    a.spg_shift = 1;
    a.err = nullptr;
    la.xs = xs;
    la.ldxs = ldxs;
    la.xs_out = xs_out;
    int rc;
    if (use_ld()) {
        // one wave of the workgroup is the loader: at most 15 consumers (4 x 4 plans run as 3 x 4)
        GemmPlan lp = pl;
        if (lp.TN * lp.WK > 15) lp.TN = 3;
        dim3 lgrid((unsigned)cdiv64(p.NT, lp.TN), (unsigned)lp.S, 1);
        rc = act == 2 ? launch_ld<2>(lp, lgrid, st, la) : launch_ld<0>(lp, lgrid, st, la);
        if (rc != TGIS_OK) return rc;
        TGIS_CHECK_LAUNCH();
        if (!partial && pl.S > 1) return gptq::reduce_slabs(slabs, a.bias, a.out, a.ldo, a.M, a.N, (int)p.NT * 32, a.S, st);
        return TGIS_OK;
    }
    dim3 grid((unsigned)cdiv64(p.NT, pl.TN), (unsigned)pl.S, 1);
    if (ring_depth() == 8)
        rc = act == 2 ? launch_tw<2, 8>(pl, grid, st, la) : launch_tw<0, 8>(pl, grid, st, la);
    else
        rc = act == 2 ? launch_tw<2, 4>(pl, grid, st, la) : launch_tw<0, 4>(pl, grid, st, la);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_LAUNCH();
    if (!partial && pl.S > 1) return gptq::reduce_slabs(slabs, a.bias, a.out, a.ldo, a.M, a.N, (int)p.NT * 32, a.S, st);
    return TGIS_OK;
}

int check_lean_args(const void* x, int64_t ldx, const float* xs, int64_t ldxs, const void* prepared, int64_t M, int64_t K,
                    int64_t N, int64_t groups, int act) {
    TGIS_CHECK_ARG(x && xs && prepared, "tgis_gptq_gemm_lean: null tensor");
    TGIS_CHECK_ARG(gptq::lean_ok(M, K, N, groups, false, act),
                   "tgis_gptq_gemm_lean: needs 1 <= M <= 32, group size 128, act 0 or 2 (got M=%ld K=%ld N=%ld groups=%ld "
                   "act=%d); see tgis_gptq_lean_ok", (long)M, (long)K, (long)N, (long)groups, act);
    TGIS_CHECK_ARG(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0, "tgis_gptq_gemm_lean: x must be 16-byte aligned rows");
    TGIS_CHECK_ARG(ldxs >= K / 16 && ldxs % 2 == 0 && ((uintptr_t)xs % 16) == 0,
                   "tgis_gptq_gemm_lean: xs rows must hold K/16 pairs and be 16-byte aligned");
    return TGIS_OK;
}

}  // namespace

#ifdef TGIS_TRACE
extern "C" int tgis_debug_set_trace_lean(void* ptr) {
    long long* p = (long long*)ptr;
    TGIS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gptq::g_trace), &p, sizeof(p)));
    return TGIS_OK;
}
#endif

extern "C" int tgis_gptq_lean_status(int reset) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ld_err), sizeof(v)) != hipSuccess) return -1;
    if (reset && v) {
        const unsigned z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ld_err), &z, sizeof(z));
    }
    return (int)v;
}

extern "C" int tgis_gptq_lean_ok(int64_t M, int64_t K, int64_t N, int64_t groups, int act_order, int act) {
    return gptq::lean_ok(M, K, N, groups, act_order != 0, act) ? 1 : 0;
}

extern "C" int tgis_xsum_f16(const void* x, int64_t ldx, float* xs, int64_t ldxs, int64_t M, int64_t K, void* stream) {
    TGIS_CHECK_ARG(x && xs, "tgis_xsum_f16: null tensor");
    TGIS_CHECK_ARG(M >= 0 && K > 0 && K % 16 == 0 && ldx % 8 == 0 && ldxs >= K / 16 && ((uintptr_t)x % 16) == 0 &&
                       ((uintptr_t)xs % 8) == 0,
                   "tgis_xsum_f16: K must be a multiple of 16, rows 16-byte aligned, ldxs >= K/16");
    if (M == 0) return TGIS_OK;
    TGIS_CHECK_ARG(M <= 65535, "tgis_xsum_f16: M too large for one launch");
    hipLaunchKernelGGL(xsum_kernel, dim3((unsigned)cdiv64(K / 8, 256), (unsigned)M), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)x, ldx, xs, ldxs, (int)K);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int tgis_gptq_gemm_f16_lean(const void* x, int64_t ldx, const float* xs, int64_t ldxs, const void* prepared,
                                       const void* bias, void* out, int64_t ldo, float* xs_out, int64_t M, int64_t K,
                                       int64_t N, int64_t groups, int act, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
    int rc = check_lean_args(x, ldx, xs, ldxs, prepared, M, K, N, groups, act);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(out, "tgis_gptq_gemm_f16_lean: null out");
    TGIS_CHECK_ARG(!xs_out || act == 2, "tgis_gptq_gemm_f16_lean: xs_out is produced by the act=2 epilogue only");
    GemmPlan pl = gptq::plan_gemm(K, N, act, M);
    const int64_t need = 4096 + gptq::slab_bytes(M, N, pl.S);
    TGIS_CHECK_ARG(workspace && workspace_bytes >= need, "tgis_gptq_gemm_f16_lean: workspace too small (%ld < %ld)",
                   (long)workspace_bytes, (long)need);
    hipStream_t st = (hipStream_t)stream;
    TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
    return launch_lean(x, ldx, xs, ldxs, prepared, bias, out, ldo, xs_out, M, K, N, groups, act,
                       (float*)((uint8_t*)workspace + 4096), 0, pl, st);
}

extern "C" int tgis_gptq_gemm_f16_partial_lean(const void* x, int64_t ldx, const float* xs, int64_t ldxs,
                                               const void* prepared, int64_t M, int64_t K, int64_t N, int64_t groups,
                                               float* slabs, int64_t slabs_bytes, int* num_slabs, int64_t* slab_ld,
                                               void* stream) {
    int rc = check_lean_args(x, ldx, xs, ldxs, prepared, M, K, N, groups, 0);
    if (rc != TGIS_OK) return rc;
    GemmPlan pl = gptq::plan_gemm(K, N, 0, M);
    TGIS_CHECK_ARG(slabs && slabs_bytes >= tgis_gptq_gemm_partial_bytes(M, K, N),
                   "tgis_gptq_gemm_f16_partial_lean: slab buffer too small");
    if (num_slabs) *num_slabs = pl.S;
    if (slab_ld) *slab_ld = cdiv64(N, 32) * 32;
    hipStream_t st = (hipStream_t)stream;
    TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
    return launch_lean(x, ldx, xs, ldxs, prepared, nullptr, nullptr, 0, nullptr, M, K, N, groups, 0, slabs, 1, pl, st);
}
