// Dense skinny GEMM for decode-sized M (f16 / bf16 weights), gfx950 MFMA 32x32x16.
// Replaces the cuBLAS calls the reference makes through F.linear / torch.mm at decode time:
// FastLinear (utils/layers.py:110-111) and the lm_head matmul (utils/layers.py:261).
//
// Prepared image: W[N,K] (torch Linear layout) repacked into 32-column MFMA tiles so that every
// wave load is one contiguous KiB:  [NT=ceil(N/32)][KS=ceil(K/64)][4][64 lanes][8 elems];
// lane l, word i = W[n = nt*32 + (l&31)][k = (ks*8 + (l>>5)*4 + i)*8 .. +7].
#include <algorithm>
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
    const void* prep;
    const void* bias;
    void* out;
    int64_t ldo;
    int M, K, N, KB, S, NT, KS;
    int out_f32;
    float* slabs;
};

constexpr int DMAXSTEPS = 4;
constexpr int DTHREADS = 512;

template <typename T, int WN, int ACT>
__global__ __launch_bounds__(DTHREADS) void dense_gemm_kernel(DenseArgs a) {
    using V8 = typename VecT<T>::x8;
    constexpr int WK = 8 / WN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wn = w % WN, wk = w / WN;
    const int ntg = blockIdx.x, split = blockIdx.y;
    const int kb0 = split * a.KB;
    const int kb1 = min(a.K, kb0 + a.KB);
    const int steps_total = (kb1 - kb0 + 63) >> 6;
    const int rs = a.KB + 8;
    T* xs = reinterpret_cast<T*>(smem);
    const T* x = reinterpret_cast<const T*>(a.x);

    const int spw = (steps_total + WK - 1) / WK;
    const int st0 = wk * spw;
    const int nt = ntg * WN + wn;
    const int nsteps = (nt < a.NT) ? max(0, min(spw, steps_total - st0)) : 0;
    const int ks0 = (kb0 >> 6) + st0;
    V8 wv[DMAXSTEPS][4];
    const V8* wbase = reinterpret_cast<const V8*>(a.prep) + ((int64_t)nt * a.KS + ks0) * 256 + lane;
#pragma unroll
    for (int s = 0; s < DMAXSTEPS; ++s)
        if (s < nsteps) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[s][i] = __builtin_nontemporal_load(wbase + s * 256 + i * 64);
        }

    {
        const int c8n = a.KB >> 3;
        for (int idx = tid; idx < 32 * c8n; idx += DTHREADS) {
            int row = idx / c8n, c8 = idx - row * c8n;
            int k = kb0 + c8 * 8;
            V8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (T)0.f;
            if (row < a.M && k < kb1) {
                const T* xr = x + (int64_t)row * a.ldx;
                if (k + 8 <= kb1) {
                    v = ld16<V8>(xr + k);
                    if (ACT == 1) {
                        V8 u = ld16<V8>(xr + a.K + k);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float g = to_f32(v[e]);
                            float sl = g / (1.f + __expf(-g));
                            v[e] = from_f32<T>(to_f32(from_f32<T>(sl)) * to_f32(u[e]));
                        }
                    }
                } else {
                    for (int e = 0; e < 8; ++e)
                        if (k + e < kb1) {
                            float g = to_f32(xr[k + e]);
                            if (ACT == 1) {
                                float sl = g / (1.f + __expf(-g));
                                g = to_f32(from_f32<T>(sl)) * to_f32(xr[a.K + k + e]);
                            }
                            v[e] = from_f32<T>(g);
                        }
                }
            }
            st16(xs + row * rs + c8 * 8, v);
        }
    }
    __syncthreads();

    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const T* xrow = xs + (lane & 31) * rs + (lane >> 5) * 32;
#pragma unroll
    for (int s = 0; s < DMAXSTEPS; ++s)
        if (s < nsteps) {
            const T* xk = xrow + (st0 + s) * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                V8 av = ld16<V8>(xk + i * 8);
                acc = mfma32(av, wv[s][i], acc);
            }
        }

    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    {
        float* dst = red + ((wk * WN + wn) << 10);
        const int col = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            dst[row * 32 + col] = acc[r];
        }
    }
    __syncthreads();
    auto emit = [&](int m, int n, f32x4 v) {
        for (int e = 0; e < 4; ++e) {
            if (n + e < a.N) {
                float f = v[e];
                if (a.bias) f += to_f32(reinterpret_cast<const T*>(a.bias)[n + e]);
                if (a.out_f32)
                    reinterpret_cast<float*>(a.out)[(int64_t)m * a.ldo + n + e] = f;
                else
                    reinterpret_cast<T*>(a.out)[(int64_t)m * a.ldo + n + e] = from_f32<T>(f);
            }
        }
    };
    for (int o = tid; o < WN * 256; o += DTHREADS) {
        int wn2 = o >> 8, m = (o >> 3) & 31, c4 = (o & 7) * 4;
        int nt2 = ntg * WN + wn2;
        if (nt2 >= a.NT) continue;
        f32x4 v = {0, 0, 0, 0};
#pragma unroll
        for (int k2 = 0; k2 < WK; ++k2)
            v += *reinterpret_cast<const f32x4*>(red + ((k2 * WN + wn2) << 10) + m * 32 + c4);
        if (a.S == 1) {
            if (m < a.M) emit(m, nt2 * 32 + c4, v);
        } else {
            *reinterpret_cast<f32x4*>(a.slabs + (((int64_t)split * a.NT + nt2) << 10) + m * 32 + c4) = v;
        }
    }
}

// Sum the S split-K slabs in fixed order and emit the output (+bias); thread = (row, 4 columns).
template <typename T>
__global__ __launch_bounds__(256) void dense_splitk_reduce_kernel(DenseArgs a) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)a.NT * 256) return;
    int c4 = (idx & 7) * 4, m = (idx >> 3) & 31;
    int64_t nt = idx >> 8;
    if (m >= a.M) return;
    f32x4 v = {0, 0, 0, 0};
    for (int s2 = 0; s2 < a.S; ++s2)
        v += *reinterpret_cast<const f32x4*>(a.slabs + (((int64_t)s2 * a.NT + nt) << 10) + m * 32 + c4);
    int64_t n = nt * 32 + c4;
    for (int e = 0; e < 4; ++e) {
        if (n + e < a.N) {
            float f = v[e];
            if (a.bias) f += to_f32(reinterpret_cast<const T*>(a.bias)[n + e]);
            if (a.out_f32)
                reinterpret_cast<float*>(a.out)[(int64_t)m * a.ldo + n + e] = f;
            else
                reinterpret_cast<T*>(a.out)[(int64_t)m * a.ldo + n + e] = from_f32<T>(f);
        }
    }
}

struct DensePlan {
    int WN, KB, S;
    size_t lds;
};

static DensePlan plan_dense(int64_t K, int64_t N) {
    int64_t NT = cdiv64(N, 32);
    DensePlan best = {1, 1024, 1, 0};
    double best_cost = 1e30;
    const int wns[4] = {1, 2, 4, 8};
    for (int wi = 0; wi < 4; ++wi) {
        int WN = wns[wi], WK = 8 / WN;
        int64_t kbmax = std::min<int64_t>(2048, (int64_t)DMAXSTEPS * 64 * WK);
        for (int64_t S = 1; S <= 64; ++S) {
            int64_t KB = cdiv64(cdiv64(K, S), 64) * 64;
            if (KB > kbmax) continue;
            if ((S - 1) * KB >= K) continue;
            int64_t blocks = cdiv64(NT, WN) * S;
            size_t lds = std::max<size_t>(32 * (KB + 8) * 2, 8 * 4096);
            int per_cu = std::min<int>(4, (int)(160 * 1024 / (lds + 64)));
            if (per_cu < 1) continue;
            double rounds = (double)blocks / (256.0 * per_cu);
            double fill = rounds < 1.0 ? 1.0 : (std::ceil(rounds) / rounds);
            double wbytes = (double)K * N * 2;
            double xbytes = (double)blocks * 32 * KB * 2 * 0.25;
            double sbytes = S > 1 ? (double)S * 32 * N * 4 * 2.0 : 0.0;
            double under = blocks < 256 ? 256.0 / blocks : 1.0;
            double cost = (wbytes + xbytes + sbytes) * fill * under;
            if (cost < best_cost) {
                best_cost = cost;
                best = {WN, (int)KB, (int)S, lds};
            }
        }
    }
    return best;
}

template <typename T, int WN, int ACT>
static int launch_dense(const DenseArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        TGIS_CHECK_HIP(hipFuncSetAttribute((const void*)dense_gemm_kernel<T, WN, ACT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL((dense_gemm_kernel<T, WN, ACT>), grid, dim3(DTHREADS), lds, st, a);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

template <typename T>
static int dispatch_dense(const DenseArgs& a, int WN, int act, dim3 grid, size_t lds, hipStream_t st) {
    switch (WN * 2 + act) {
        case 2: return launch_dense<T, 1, 0>(a, grid, lds, st);
        case 3: return launch_dense<T, 1, 1>(a, grid, lds, st);
        case 4: return launch_dense<T, 2, 0>(a, grid, lds, st);
        case 5: return launch_dense<T, 2, 1>(a, grid, lds, st);
        case 8: return launch_dense<T, 4, 0>(a, grid, lds, st);
        case 9: return launch_dense<T, 4, 1>(a, grid, lds, st);
        case 16: return launch_dense<T, 8, 0>(a, grid, lds, st);
        case 17: return launch_dense<T, 8, 1>(a, grid, lds, st);
    }
    tgis_set_error("dispatch_dense: bad WN/act");
    return TGIS_EINVAL;
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
    (void)M;
    DensePlan pl = plan_dense(K, N);
    return 4096 + (pl.S > 1 ? (int64_t)pl.S * cdiv64(N, 32) * 4096 : 0);
}

extern "C" int tgis_dense_gemm(const void* x, int64_t ldx, const void* prepared, const void* bias, void* out,
                               int64_t ldo, int64_t M, int64_t K, int64_t N, int dtype, int out_f32, int act,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    TGIS_CHECK_ARG(x && prepared && out, "tgis_dense_gemm: null tensor");
    TGIS_CHECK_ARG(M >= 0 && K > 0 && N > 0, "tgis_dense_gemm: bad shape");
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_dense_gemm: bad dtype");
    TGIS_CHECK_ARG(act == 0 || act == 1, "tgis_dense_gemm: act must be 0 or 1");
    TGIS_CHECK_ARG(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0, "tgis_dense_gemm: x rows must be 16-byte aligned");
    if (M == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    DensePlan pl = plan_dense(K, N);
    int64_t NT = cdiv64(N, 32), KS = cdiv64(K, 64);
    int64_t need = 4096 + (pl.S > 1 ? (int64_t)pl.S * NT * 4096 : 0);
    TGIS_CHECK_ARG(workspace && workspace_bytes >= need, "tgis_dense_gemm: workspace too small (%ld < %ld)",
                   (long)workspace_bytes, (long)need);
    TgisTimedScope timed(TGIS_OP_DENSE_GEMM, st);
    DenseArgs a;
    a.prep = prepared;
    a.bias = bias;
    a.ldx = ldx;
    a.ldo = ldo;
    a.K = (int)K;
    a.N = (int)N;
    a.KB = pl.KB;
    a.S = pl.S;
    a.NT = (int)NT;
    a.KS = (int)KS;
    a.out_f32 = out_f32;
    a.slabs = (float*)((uint8_t*)workspace + 4096);
    dim3 grid((unsigned)cdiv64(NT, pl.WN), (unsigned)pl.S);
    const int64_t esz_out = out_f32 ? 4 : 2;
    for (int64_t m0 = 0; m0 < M; m0 += 32) {
        a.x = (const uint8_t*)x + m0 * ldx * 2;
        a.out = (uint8_t*)out + m0 * ldo * esz_out;
        a.M = (int)std::min<int64_t>(32, M - m0);
        int rc = dtype == TGIS_F16 ? dispatch_dense<f16>(a, pl.WN, act, grid, pl.lds, st)
                                   : dispatch_dense<bf16>(a, pl.WN, act, grid, pl.lds, st);
        if (rc != TGIS_OK) return rc;
        if (pl.S > 1) {
            if (dtype == TGIS_F16)
                hipLaunchKernelGGL(dense_splitk_reduce_kernel<f16>, dim3((unsigned)NT), dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL(dense_splitk_reduce_kernel<bf16>, dim3((unsigned)NT), dim3(256), 0, st, a);
            TGIS_CHECK_LAUNCH();
        }
    }
    return TGIS_OK;
}
