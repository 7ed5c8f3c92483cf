// experiments/README.md: add + RMSNorm as the first phase of the GEMM behind it (grid barrier) — built, verified, measured
// 1 - 1.5 us per layer SLOWER than the separate norm launches; not part of libtgis_hip.so.
// This translation unit IS the product's csrc/gptq.hip (included whole, so that its file-local launchers are visible) plus
// the experimental entry points of include/tgis_experiments.h; experiments/build.py compiles it INSTEAD of the product file.
#include "../include/tgis_experiments.h"
#include "gptq.hip"

// ---- add + RMSNorm as the first phase of the GEMM behind it -----------------------------------------------------------
namespace {
std::mutex g_gemm_bar_mu;
gsync::GridBar* g_gemm_bar[16] = {};

// The grid barrier of the two-phase launches, one per device, library-owned; allocated by the first call that is not inside
// a stream capture (tgis_norm_gemm_ok does it too).  Launches of one device that use it must not overlap (one stream).
gsync::GridBar* gemm_bar(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lock(g_gemm_bar_mu);
    if (!g_gemm_bar[dev]) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (st && (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)) {
            (void)hipGetLastError();
            return nullptr;
        }
        gsync::GridBar* b = nullptr;
        if (hipMalloc((void**)&b, sizeof(gsync::GridBar)) != hipSuccess ||
            hipMemset(b, 0, sizeof(gsync::GridBar)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        (void)hipDeviceSynchronize();
        g_gemm_bar[dev] = b;
    }
    return g_gemm_bar[dev];
}

int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
            cus = v;
    }
    return cus;
}

// shape conditions of the two-phase launch; *blocks receives the grid size
bool norm_gemm_shape_ok(int64_t M, int64_t K, int64_t N, int64_t groups, int act_order, int act, int64_t* blocks) {
    if (M < 1 || M > 32 || act_order || groups <= 0 || K % groups || (act != 2 && act != 3)) return false;
    const int64_t gs = K / groups, spg = gs / 64;
    if (!(groups == 1 || (gs % 64 == 0 && (spg & (spg - 1)) == 0))) return false;
    if (K % 8 || K > 16384) return false;  // K is the hidden size the norm phase normalises
    const GemmPlan pl = plan_gemm(K, N, 2, M);
    const int64_t nb = cdiv64(cdiv64(N, 32), pl.TN);
    if (blocks) *blocks = nb;
    // every workgroup must be resident for the grid barrier (one per CU: the x buffers fill the LDS), and row r is
    // normalised by workgroup r
    return pl.S == 1 && nb >= M && nb <= device_cus();
}


int norm_phase_of(const tgis_norm_in* n, int64_t M, int64_t K, gsync::NormPhase* p) {
    TGIS_CHECK_ARG(n && (n->slabs || n->x) && n->weight && n->y, "tgis_gptq_norm_gemm: null norm tensor");
    TGIS_CHECK_ARG(!n->slabs || (n->num_slabs >= 1 && n->slab_ld >= K && n->slab_ld % 4 == 0),
                   "tgis_gptq_norm_gemm: partial input needs a slab row stride >= hidden");
    p->y_frag = 0;
    p->slabs = n->slabs;
    p->S = n->num_slabs;
    p->slab_ld = n->slab_ld;
    p->xbias = n->slabs ? n->slab_bias : nullptr;
    p->x = n->x;
    p->residual = n->residual;
    p->weight = n->weight;
    p->y = n->y;
    p->res_out = n->res_out;
    p->rows = (int)M;
    p->hidden = (int)K;
    p->eps = n->eps;
    return TGIS_OK;
}
}  // namespace

extern "C" int tgis_gptq_norm_gemm_ok(int64_t M, int64_t K, int64_t N, int64_t groups, int act_order, int act) {
    if (getenv("TGIS_ALLOW_SHARED_GPU")) return 0;  // several processes on one GPU: residency of a whole grid is not ours to assume
    if (!norm_gemm_shape_ok(M, K, N, groups, act_order, act, nullptr)) return 0;
    return gemm_bar(nullptr) ? 1 : 0;
}

extern "C" int tgis_gptq_norm_gemm_status(int reset) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || !g_gemm_bar[dev]) return 0;
    unsigned err = 0;
    if (hipMemcpy(&err, &g_gemm_bar[dev]->err, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (err && reset) (void)hipMemset(g_gemm_bar[dev], 0, sizeof(gsync::GridBar));
    return (int)err;
}

extern "C" int tgis_gptq_norm_gate_up_f16(const tgis_norm_in* norm, const void* prepared, const void* bias, void* out,
                                          int64_t ldo, int64_t M, int64_t K, int64_t N, int64_t groups, void* stream) {
    gsync::NormPhase np;
    int rc = norm_phase_of(norm, M, K, &np);
    if (rc != TGIS_OK) return rc;
    rc = check_gemm_args(np.y, K, prepared, M, K, N, groups, 2);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(out, "tgis_gptq_norm_gate_up_f16: null out");
    int64_t blocks = 0;
    TGIS_CHECK_ARG(norm_gemm_shape_ok(M, K, N, groups, 0, 2, &blocks), "tgis_gptq_norm_gate_up_f16: shape not served by the "
                   "two-phase launch (M=%ld K=%ld N=%ld groups=%ld); see tgis_gptq_norm_gemm_ok", (long)M, (long)K, (long)N,
                   (long)groups);
    hipStream_t st = (hipStream_t)stream;
    gsync::GridBar* bar = gemm_bar(st);
    TGIS_CHECK_ARG(bar, "tgis_gptq_norm_gate_up_f16: the grid barrier is allocated by the first call outside a stream capture");
    GemmPlan pl = plan_gemm(K, N, 2, M);
    TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
    return launch_gptq(np.y, K, prepared, bias, nullptr, out, ldo, M, K, N, groups, 2, nullptr, 0, pl, st, nullptr, &np, bar);
}

extern "C" int tgis_gptq_norm_qkv_rope_f16(const tgis_norm_in* norm, const void* prepared, const void* bias,
                                           const int32_t* positions, const int32_t* slots, const void* cos, const void* sin,
                                           void* q_out, int64_t ldq, void* k_pool, void* v_pool, int64_t M, int64_t K,
                                           int64_t N, int64_t groups, int64_t H, int64_t Hkv, int64_t D, void* stream) {
    gsync::NormPhase np;
    int rc = norm_phase_of(norm, M, K, &np);
    if (rc != TGIS_OK) return rc;
    rc = check_gemm_args(np.y, K, prepared, M, K, N, groups, 0);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(positions && slots && cos && sin && q_out && k_pool && v_pool, "tgis_gptq_norm_qkv_rope_f16: null tensor");
    TGIS_CHECK_ARG(D >= 32 && D % 32 == 0 && H >= 1 && Hkv >= 1 && (H + 2 * Hkv) * D == N && ldq >= H * D,
                   "tgis_gptq_norm_qkv_rope_f16: N must be (H + 2 Hkv) * D, D a multiple of 32");
    int64_t blocks = 0;
    TGIS_CHECK_ARG(norm_gemm_shape_ok(M, K, N, groups, 0, 3, &blocks), "tgis_gptq_norm_qkv_rope_f16: shape not served by the "
                   "two-phase launch (M=%ld K=%ld N=%ld groups=%ld); see tgis_gptq_norm_gemm_ok", (long)M, (long)K, (long)N,
                   (long)groups);
    hipStream_t st = (hipStream_t)stream;
    gsync::GridBar* bar = gemm_bar(st);
    TGIS_CHECK_ARG(bar, "tgis_gptq_norm_qkv_rope_f16: the grid barrier is allocated by the first call outside a stream capture");
    GemmPlan pl = plan_gemm(K, N, 2, M);
    RopeEpi rope{positions, slots, (const f16*)cos, (const f16*)sin, (f16*)k_pool, (f16*)v_pool, (int)H, (int)Hkv, (int)D};
    TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
    return launch_gptq(np.y, K, prepared, bias, nullptr, q_out, ldq, M, K, N, groups, 3, nullptr, 0, pl, st, &rope, &np, bar);
}

