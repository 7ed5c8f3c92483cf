// GPTQ int4 linear for gfx950: one-time repack ("prepare"), fused dequant + MFMA GEMM for decode-sized
// M, and a full dequant kernel for the large-M (library GEMM) path.
//
// Replaces exllamav2_kernels.make_q_matrix / gemm_half_q_half as called from
// utils/gptq/exllamav2.py:14-62,124-144.  Normative arithmetic (utils/gptq/quant_linear.py:130-138,
// 184-194):  W[k,n] = (q[k,n] - (z[g(k),n] + 1)) * s[g(k),n];  y = x @ W, fp32 accumulate, f16 out.
//
// Prepared image, NT = ceil(N/32) column tiles, KS = ceil(K/64) k-steps, G groups:
//   A: wq  [NT][KS][64 lanes][4] int32 — lane l word i = the 8 nibbles of rows
//         k = (ks*8 + (l>>5)*4 + i)*8 .. +7 of column n = nt*32 + (l&31)   (1 KiB per wave load,
//         and exactly the B-operand fragment of v_mfma_f32_32x32x16_f16: 8 consecutive k per lane)
//   B: scl [NT][G][32] f16,   C: zp1 [NT][G][32] u8 (= z + 1, 1..16)
// Rows are pre-permuted by the act-order permutation when g_idx is not trivial.
#include <algorithm>
#include <numeric>
#include <vector>
#include "common.h"

namespace {

struct PrepLayout {
    int64_t NT, KS, G, offB, offC, total;
};
static PrepLayout prep_layout(int64_t K, int64_t N, int64_t G) {
    PrepLayout p;
    p.NT = cdiv64(N, 32);
    p.KS = cdiv64(K, 64);
    p.G = G;
    p.offB = p.NT * p.KS * 1024;
    p.offC = p.offB + p.NT * G * 64;
    p.total = p.offC + p.NT * G * 32;
    p.total = (p.total + 255) & ~int64_t(255);
    return p;
}

__global__ void gptq_prepare_w_kernel(const int32_t* __restrict__ qweight, const int32_t* __restrict__ perm,
                                      int32_t* __restrict__ wq, int64_t K, int64_t N, int64_t NT, int64_t KS) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = NT * KS * 256;
    if (idx >= total) return;
    int i = idx & 3;
    int l = (idx >> 2) & 63;
    int64_t ks = (idx >> 8) % KS;
    int64_t nt = (idx >> 8) / KS;
    int64_t n = nt * 32 + (l & 31);
    int64_t p = ks * 8 + (l >> 5) * 4 + i;  // k-pack row (8 k each)
    uint32_t v = 0;
    if (n < N && p * 8 < K) {
        if (perm == nullptr) {
            v = (uint32_t)qweight[p * N + n];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int64_t ksrc = perm[p * 8 + e];
                uint32_t w = (uint32_t)qweight[(ksrc >> 3) * N + n];
                v |= ((w >> (4 * (ksrc & 7))) & 15u) << (4 * e);
            }
        }
    }
    wq[idx] = (int32_t)v;
}

__global__ void gptq_prepare_sz_kernel(const int32_t* __restrict__ qzeros, const f16* __restrict__ scales,
                                       f16* __restrict__ scl, uint8_t* __restrict__ zp1, int64_t N,
                                       int64_t NT, int64_t G) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NT * G * 32) return;
    int c = idx & 31;
    int64_t g = (idx >> 5) % G;
    int64_t nt = (idx >> 5) / G;
    int64_t n = nt * 32 + c;
    f16 s = (f16)0.f;
    uint8_t z = 1;
    if (n < N) {
        s = scales[g * N + n];
        uint32_t w = (uint32_t)qzeros[g * (N / 8) + (n >> 3)];
        z = (uint8_t)(((w >> (4 * (n & 7))) & 15u) + 1u);
    }
    scl[idx] = s;
    zp1[idx] = z;
}

// 8 nibbles of q -> 8 halves (q_e - zp1) in the order [0,4,1,5,2,6,3,7] (exact integer arithmetic).
__device__ __forceinline__ f16x8 dequant8(uint32_t q, f16x2 zc, f16x2 zd) {
    const uint32_t M0 = 0x000F000Fu, M1 = 0x00F000F0u, EX = 0x64006400u;  // 0x6400 = 1024.0h
    const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
    uint32_t q2 = q >> 8;
    uint32_t a0 = (q & M0) | EX;   // 1024 + n0 , 1024 + n4
    uint32_t a1 = (q & M1) | EX;   // 1024 + 16 n1 , 1024 + 16 n5
    uint32_t a2 = (q2 & M0) | EX;  // n2, n6
    uint32_t a3 = (q2 & M1) | EX;  // n3, n7
    f16x2 h0 = __builtin_bit_cast(f16x2, a0) - zc;
    f16x2 h1 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a1), r16, zd);
    f16x2 h2 = __builtin_bit_cast(f16x2, a2) - zc;
    f16x2 h3 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a3), r16, zd);
    f16x8 r;
    r[0] = h0[0]; r[1] = h0[1]; r[2] = h1[0]; r[3] = h1[1];
    r[4] = h2[0]; r[5] = h2[1]; r[6] = h3[0]; r[7] = h3[1];
    return r;
}

struct GemmArgs {
    const f16* x;
    int64_t ldx;
    const uint8_t* prep;
    int64_t offB, offC;
    const f16* bias;
    const int32_t* perm;
    f16* out;
    int64_t ldo;
    int M, K, N;        // M = rows in this slab (<=32)
    int G, gs;          // groups, group size
    int KB;             // k-range per block (multiple of 64)
    int S;              // global k splits
    int NT, KS;
    float* slabs;       // [S][NT][32*32] f32
    unsigned* counters; // [ceil(NT/WN)]
};

constexpr int MAXSTEPS = 8;  // k64-steps per wave
constexpr int GEMM_THREADS = 512;

// Block = 8 waves = WN column tiles x WK=8/WN k-parts over one [KB x 32*WN] rectangle of W; the
// x slab [32][KB] (f16, optionally silu(gate)*up fused) is staged once in LDS in the nibble order.
template <int WN, int ACT, bool GROUP_ACC>
__global__ __launch_bounds__(GEMM_THREADS) void gptq_gemm_kernel(GemmArgs a) {
    constexpr int WK = 8 / WN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int wn = w % WN, wk = w / WN;
    const int ntg = blockIdx.x, split = blockIdx.y;
    const int kb0 = split * a.KB;
    const int kb1 = min(a.K, kb0 + a.KB);
    const int klen = kb1 - kb0;                   // >0 by construction
    const int steps_total = (klen + 63) >> 6;
    const int rs = a.KB + 8;                      // LDS row stride in halves (+16 B: conflict-free b128)
    f16* xs = reinterpret_cast<f16*>(smem);

    // ---- issue this wave's weight loads first (HBM latency overlaps the x staging) ------------
    const int spw = (steps_total + WK - 1) / WK;
    const int st0 = wk * spw;
    const int nt = ntg * WN + wn;
    const bool active = nt < a.NT;
    const int nsteps = active ? max(0, min(spw, steps_total - st0)) : 0;
    const int ks0 = (kb0 >> 6) + st0;
    u32x4 wq[MAXSTEPS];
    f16 sc[MAXSTEPS];
    uint8_t zz[MAXSTEPS];
    const u32x4* wbase = reinterpret_cast<const u32x4*>(a.prep) + ((int64_t)nt * a.KS + ks0) * 64 + lane;
    const f16* sbase = reinterpret_cast<const f16*>(a.prep + a.offB) + (int64_t)nt * a.G * 32 + (lane & 31);
    const uint8_t* zbase = a.prep + a.offC + (int64_t)nt * a.G * 32 + (lane & 31);
#pragma unroll
    for (int s = 0; s < MAXSTEPS; ++s) {
        if (s < nsteps) {
            wq[s] = __builtin_nontemporal_load(wbase + s * 64);
            if (GROUP_ACC) {
                int g = ((ks0 + s) * 64) / a.gs;
                sc[s] = sbase[g * 32];
                zz[s] = zbase[g * 32];
            }
        }
    }

    // ---- stage x[0:32, kb0:kb1] into LDS (zero-padded), 16 B per thread per iteration ---------
    {
        const int c8n = a.KB >> 3;  // 16-byte chunks per row
        for (int idx = tid; idx < 32 * c8n; idx += GEMM_THREADS) {
            int row = idx / c8n, c8 = idx - row * c8n;
            int k = kb0 + c8 * 8;
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (row < a.M && k < kb1) {
                const f16* xr = a.x + (int64_t)row * a.ldx;
                if (a.perm == nullptr) {
                    v = ld16<f16x8>(xr + k);
                    if (ACT == 1) {
                        f16x8 u = ld16<f16x8>(xr + a.K + k);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float gte = (float)v[e];
                            float sl = gte / (1.f + __expf(-gte));
                            // reference rounds silu(gate) to f16 before the multiply (eager torch ops)
                            v[e] = (f16)((float)(f16)sl * (float)u[e]);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        int ksrc = a.perm[k + e];
                        float gte = (float)xr[ksrc];
                        if (ACT == 1) {
                            float sl = gte / (1.f + __expf(-gte));
                            gte = (float)(f16)sl * (float)xr[a.K + ksrc];
                        }
                        v[e] = (f16)gte;
                    }
                }
            }
            f16x8 p;  // nibble order [0,4,1,5,2,6,3,7]
            p[0] = v[0]; p[1] = v[4]; p[2] = v[1]; p[3] = v[5];
            p[4] = v[2]; p[5] = v[6]; p[6] = v[3]; p[7] = v[7];
            st16(xs + row * rs + c8 * 8, p);
        }
    }
    __syncthreads();

    // ---- dequant + MFMA ------------------------------------------------------------------------
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 accg = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const f16* xrow = xs + (lane & 31) * rs + (lane >> 5) * 32;
#pragma unroll
    for (int s = 0; s < MAXSTEPS; ++s) {
        if (s < nsteps) {
            const f16* xk = xrow + (st0 + s) * 64;
            if (GROUP_ACC) {
                float zf = (float)zz[s];
                f16 zc1 = (f16)(1024.f + zf), zd1 = (f16)(-64.f - zf);
                f16x2 zc = {zc1, zc1}, zd = {zd1, zd1};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f16x8 b = dequant8(wq[s][i], zc, zd);
                    f16x8 av = ld16<f16x8>(xk + i * 8);
                    accg = mfma32(av, b, accg);
                }
                int g = ((ks0 + s) * 64) / a.gs;
                bool last = (s + 1 == nsteps) || (((ks0 + s + 1) * 64) / a.gs != g);
                if (last) {
                    float sf = (float)sc[s];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc[r] = fmaf(sf, accg[r], acc[r]);
                        accg[r] = 0.f;
                    }
                }
            } else {
                // group size not a multiple of 64: scale each k-pack's weights in f16 (exllama-style)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int k = ((ks0 + s) * 8 + (lane >> 5) * 4 + i) * 8;
                    int g = min(k / a.gs, a.G - 1);
                    float zf = (float)zbase[g * 32];
                    f16 sv = sbase[g * 32];
                    f16 zc1 = (f16)(1024.f + zf), zd1 = (f16)(-64.f - zf);
                    f16x2 zc = {zc1, zc1}, zd = {zd1, zd1};
                    f16x8 b = dequant8(wq[s][i], zc, zd);
#pragma unroll
                    for (int e = 0; e < 8; ++e) b[e] = b[e] * sv;
                    f16x8 av = ld16<f16x8>(xk + i * 8);
                    acc = mfma32(av, b, acc);
                }
            }
        }
    }

    // ---- in-block reduce over the WK k-parts ----------------------------------------------------
    __syncthreads();  // everyone is done reading the x slab; reuse LDS as [WK][WN][32][32] f32
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
    // thread -> (tile wn2, row m, 4 consecutive columns)
    for (int o = tid; o < WN * 256; o += GEMM_THREADS) {
        int wn2 = o >> 8, m = (o >> 3) & 31, c4 = (o & 7) * 4;
        int nt2 = ntg * WN + wn2;
        if (nt2 >= a.NT) continue;
        f32x4 v = {0, 0, 0, 0};
#pragma unroll
        for (int k2 = 0; k2 < WK; ++k2) {
            f32x4 t = *reinterpret_cast<const f32x4*>(red + ((k2 * WN + wn2) << 10) + m * 32 + c4);
            v += t;
        }
        if (a.S == 1) {
            if (m < a.M) {
                int n = nt2 * 32 + c4;
                f16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float f = v[e];
                    if (a.bias && n + e < a.N) f += (float)a.bias[n + e];
                    h[e] = (f16)f;
                }
                if (n + 3 < a.N) {
                    *reinterpret_cast<f16x4*>(a.out + (int64_t)m * a.ldo + n) = h;
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (n + e < a.N) a.out[(int64_t)m * a.ldo + n + e] = h[e];
                }
            }
        } else {
            *reinterpret_cast<f32x4*>(a.slabs + (((int64_t)split * a.NT + nt2) << 10) + m * 32 + c4) = v;
        }
    }
    if (a.S == 1) return;

    // ---- cross-block split-K: last arriver sums the S slabs in fixed order (deterministic) -------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // flag lives in the dynamic LDS region (a static __shared__ would misalign its base, guide G17)
    volatile int* s_last = reinterpret_cast<volatile int*>(smem);
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned t = __hip_atomic_fetch_add(a.counters + ntg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = (t == (unsigned)(a.S - 1));
    }
    __syncthreads();
    if (!*s_last) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(a.counters + ntg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int o = tid; o < WN * 256; o += GEMM_THREADS) {
        int wn2 = o >> 8, m = (o >> 3) & 31, c4 = (o & 7) * 4;
        int nt2 = ntg * WN + wn2;
        if (nt2 >= a.NT || m >= a.M) continue;
        f32x4 v = {0, 0, 0, 0};
        for (int s2 = 0; s2 < a.S; ++s2) {
            const float* p = a.slabs + (((int64_t)s2 * a.NT + nt2) << 10) + m * 32 + c4;
            f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
            v += t;
        }
        int n = nt2 * 32 + c4;
        for (int e = 0; e < 4; ++e) {
            if (n + e < a.N) {
                float f = v[e];
                if (a.bias) f += (float)a.bias[n + e];
                a.out[(int64_t)m * a.ldo + n + e] = (f16)f;
            }
        }
    }
}

__global__ void gptq_dequant_kernel(const uint8_t* __restrict__ prep, int64_t offB, int64_t offC,
                                    f16* __restrict__ wout, int K, int N, int G, int gs, int NT, int KS) {
    // one thread per prepared int32 (8 k of one column)
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)NT * KS * 256) return;
    int i = idx & 3;
    int l = (idx >> 2) & 63;
    int64_t ks = (idx >> 8) % KS;
    int64_t nt = (idx >> 8) / KS;
    int n = nt * 32 + (l & 31);
    int k0 = (ks * 8 + (l >> 5) * 4 + i) * 8;
    if (n >= N || k0 >= K) return;
    uint32_t q = reinterpret_cast<const uint32_t*>(prep)[idx];
    int g = min(k0 / gs, G - 1);
    float s = (float)reinterpret_cast<const f16*>(prep + offB)[(nt * G + g) * 32 + (l & 31)];
    float z = (float)prep[offC + (nt * G + g) * 32 + (l & 31)];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = ((float)((q >> (4 * e)) & 15u) - z) * s;
        wout[(int64_t)(k0 + e) * N + n] = (f16)v;
    }
}

struct GemmPlan {
    int WN, KB, S;
    size_t lds;
};

static GemmPlan plan_gemm(int64_t K, int64_t N) {
    // Rectangle per block: KB x (32*WN).  Few k-splits (slab traffic = S*M*N*8 B) versus enough blocks
    // to cover 256 CUs.  LDS = 32*(KB+8)*2 bytes must leave room for >= 1 block/CU.
    int64_t NT = cdiv64(N, 32);
    GemmPlan best = {1, 1024, 1, 0};
    double best_cost = 1e30;
    const int wns[4] = {1, 2, 4, 8};
    for (int wi = 0; wi < 4; ++wi) {
        int WN = wns[wi], WK = 8 / WN;
        int64_t kbmax = std::min<int64_t>(2048, (int64_t)MAXSTEPS * 64 * WK);
        for (int64_t S = 1; S <= 64; ++S) {
            int64_t KB = cdiv64(cdiv64(K, S), 64) * 64;
            if (KB > kbmax) continue;
            if ((S - 1) * KB >= K) continue;  // empty last split
            int64_t blocks = cdiv64(NT, WN) * S;
            size_t lds = std::max<size_t>(32 * (KB + 8) * 2, 8 * 4096);
            int per_cu = std::min<int>(4, (int)(160 * 1024 / (lds + 64)));
            if (per_cu < 1) continue;
            double rounds = (double)blocks / (256.0 * per_cu);
            double fill = rounds < 1.0 ? 1.0 : (std::ceil(rounds) / rounds);
            // bytes: weights + x restaging through L2 (cheaper, x0.25) + slab round trip (x2, when S>1)
            double wbytes = (double)K * N / 2;
            double xbytes = (double)blocks * 32 * KB * 2 * 0.25;
            double sbytes = S > 1 ? (double)S * 32 * N * 4 * 2.0 : 0.0;
            double under = blocks < 256 ? 256.0 / blocks : 1.0;  // idle CUs
            double cost = (wbytes + xbytes + sbytes) * fill * under;
            if (cost < best_cost) {
                best_cost = cost;
                best = {WN, (int)KB, (int)S, lds};
            }
        }
    }
    return best;
}

template <int WN, int ACT>
static void launch_gemm(const GemmArgs& a, bool group_acc, dim3 grid, size_t lds, hipStream_t st) {
    if (group_acc)
        hipLaunchKernelGGL((gptq_gemm_kernel<WN, ACT, true>), grid, dim3(GEMM_THREADS), lds, st, a);
    else
        hipLaunchKernelGGL((gptq_gemm_kernel<WN, ACT, false>), grid, dim3(GEMM_THREADS), lds, st, a);
}

template <int WN, int ACT>
static hipError_t set_lds_attr(size_t lds) {
    hipError_t e = hipFuncSetAttribute((const void*)gptq_gemm_kernel<WN, ACT, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)gptq_gemm_kernel<WN, ACT, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace

extern "C" int64_t tgis_gptq_prepared_bytes(int64_t K, int64_t N, int64_t groups) {
    if (K <= 0 || N <= 0 || groups <= 0) return 0;
    return prep_layout(K, N, groups).total;
}

extern "C" int tgis_gptq_prepare(const int32_t* qweight, const int32_t* qzeros, const void* scales,
                                 const int32_t* g_idx_host, int32_t* perm_out, int64_t K, int64_t N,
                                 int64_t groups, void* prepared, void* stream) {
    TGIS_CHECK_ARG(qweight && qzeros && scales && prepared, "tgis_gptq_prepare: null tensor");
    TGIS_CHECK_ARG(K > 0 && N > 0 && K % 32 == 0 && N % 32 == 0,
                   "tgis_gptq_prepare: K (%ld) and N (%ld) must be positive multiples of 32", (long)K, (long)N);
    TGIS_CHECK_ARG(groups > 0 && K % groups == 0, "tgis_gptq_prepare: K %% groups != 0");
    const int64_t gs = K / groups;
    TGIS_CHECK_ARG(gs % 8 == 0, "tgis_gptq_prepare: group size %ld not a multiple of 8", (long)gs);
    hipStream_t st = (hipStream_t)stream;
    const int32_t* perm_dev = nullptr;
    if (g_idx_host) {
        bool trivial = true;
        for (int64_t k = 0; k < K; ++k)
            if (g_idx_host[k] != (int32_t)(k / gs)) { trivial = false; break; }
        if (!trivial) {
            TGIS_CHECK_ARG(perm_out, "tgis_gptq_prepare: act-order g_idx needs perm_out");
            std::vector<int32_t> perm(K);
            std::iota(perm.begin(), perm.end(), 0);
            std::stable_sort(perm.begin(), perm.end(),
                             [&](int32_t a, int32_t b) { return g_idx_host[a] < g_idx_host[b]; });
            // every group must own exactly gs rows (true for GPTQ act-order checkpoints)
            for (int64_t k = 0; k < K; ++k)
                TGIS_CHECK_ARG(g_idx_host[perm[k]] == (int32_t)(k / gs),
                               "tgis_gptq_prepare: g_idx groups are not of uniform size");
            TGIS_CHECK_HIP(hipStreamSynchronize(st));
            TGIS_CHECK_HIP(hipMemcpy(perm_out, perm.data(), K * sizeof(int32_t), hipMemcpyHostToDevice));
            perm_dev = perm_out;
        }
    }
    PrepLayout p = prep_layout(K, N, groups);
    uint8_t* base = (uint8_t*)prepared;
    int64_t totalA = p.NT * p.KS * 256;
    hipLaunchKernelGGL(gptq_prepare_w_kernel, dim3((unsigned)cdiv64(totalA, 256)), dim3(256), 0, st, qweight,
                       perm_dev, (int32_t*)base, K, N, p.NT, p.KS);
    TGIS_CHECK_LAUNCH();
    int64_t totalB = p.NT * groups * 32;
    hipLaunchKernelGGL(gptq_prepare_sz_kernel, dim3((unsigned)cdiv64(totalB, 256)), dim3(256), 0, st, qzeros,
                       (const f16*)scales, (f16*)(base + p.offB), base + p.offC, N, p.NT, groups);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

extern "C" int64_t tgis_gptq_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N) {
    (void)M;
    GemmPlan pl = plan_gemm(K, N);
    int64_t NT = cdiv64(N, 32);
    return 4096 + (pl.S > 1 ? (int64_t)pl.S * NT * 4096 : 0);
}

extern "C" int tgis_gptq_gemm_f16(const void* x, int64_t ldx, const void* prepared, const void* bias,
                                  const int32_t* perm, void* out, int64_t ldo, int64_t M, int64_t K,
                                  int64_t N, int64_t groups, int act, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    TGIS_CHECK_ARG(x && prepared && out, "tgis_gptq_gemm_f16: null tensor");
    TGIS_CHECK_ARG(M >= 0 && K > 0 && N > 0 && K % 32 == 0 && N % 32 == 0, "tgis_gptq_gemm_f16: bad shape");
    TGIS_CHECK_ARG(groups > 0 && K % groups == 0, "tgis_gptq_gemm_f16: K %% groups != 0");
    TGIS_CHECK_ARG(act == 0 || act == 1, "tgis_gptq_gemm_f16: act must be 0 or 1");
    TGIS_CHECK_ARG(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0, "tgis_gptq_gemm_f16: x must be 16-byte aligned rows");
    TGIS_CHECK_ARG(ldo % 4 == 0 && ((uintptr_t)out % 8) == 0, "tgis_gptq_gemm_f16: out rows must be 8-byte aligned");
    if (M == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    PrepLayout p = prep_layout(K, N, groups);
    GemmPlan pl = plan_gemm(K, N);
    int64_t need = 4096 + (pl.S > 1 ? (int64_t)pl.S * p.NT * 4096 : 0);
    TGIS_CHECK_ARG(workspace && workspace_bytes >= need, "tgis_gptq_gemm_f16: workspace too small (%ld < %ld)",
                   (long)workspace_bytes, (long)need);
    TGIS_CHECK_ARG(cdiv64(p.NT, pl.WN) <= 1024, "tgis_gptq_gemm_f16: N too large for the counter region");
    const int64_t gs = K / groups;
    const bool group_acc = (gs % 64 == 0) || groups == 1;

    static bool attr_done[4][2] = {};
    int wi = pl.WN == 1 ? 0 : pl.WN == 2 ? 1 : pl.WN == 4 ? 2 : 3;
    if (!attr_done[wi][act]) {
        hipError_t e = hipSuccess;
        size_t mx = 150 * 1024;
        switch (pl.WN * 2 + act) {
            case 2: e = set_lds_attr<1, 0>(mx); break;
            case 3: e = set_lds_attr<1, 1>(mx); break;
            case 4: e = set_lds_attr<2, 0>(mx); break;
            case 5: e = set_lds_attr<2, 1>(mx); break;
            case 8: e = set_lds_attr<4, 0>(mx); break;
            case 9: e = set_lds_attr<4, 1>(mx); break;
            case 16: e = set_lds_attr<8, 0>(mx); break;
            case 17: e = set_lds_attr<8, 1>(mx); break;
        }
        TGIS_CHECK_HIP(e);
        attr_done[wi][act] = true;
    }

    TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
    GemmArgs a;
    a.prep = (const uint8_t*)prepared;
    a.offB = p.offB;
    a.offC = p.offC;
    a.bias = (const f16*)bias;
    a.perm = perm;
    a.ldx = ldx;
    a.ldo = ldo;
    a.K = (int)K;
    a.N = (int)N;
    a.G = (int)groups;
    a.gs = (int)gs;
    a.KB = pl.KB;
    a.S = pl.S;
    a.NT = (int)p.NT;
    a.KS = (int)p.KS;
    a.counters = (unsigned*)workspace;
    a.slabs = (float*)((uint8_t*)workspace + 4096);
    dim3 grid((unsigned)cdiv64(p.NT, pl.WN), (unsigned)pl.S);
    for (int64_t m0 = 0; m0 < M; m0 += 32) {
        a.x = (const f16*)x + m0 * ldx;
        a.out = (f16*)out + m0 * ldo;
        a.M = (int)std::min<int64_t>(32, M - m0);
        switch (pl.WN * 2 + act) {
            case 2: launch_gemm<1, 0>(a, group_acc, grid, pl.lds, st); break;
            case 3: launch_gemm<1, 1>(a, group_acc, grid, pl.lds, st); break;
            case 4: launch_gemm<2, 0>(a, group_acc, grid, pl.lds, st); break;
            case 5: launch_gemm<2, 1>(a, group_acc, grid, pl.lds, st); break;
            case 8: launch_gemm<4, 0>(a, group_acc, grid, pl.lds, st); break;
            case 9: launch_gemm<4, 1>(a, group_acc, grid, pl.lds, st); break;
            case 16: launch_gemm<8, 0>(a, group_acc, grid, pl.lds, st); break;
            case 17: launch_gemm<8, 1>(a, group_acc, grid, pl.lds, st); break;
        }
        TGIS_CHECK_LAUNCH();
    }
    return TGIS_OK;
}

extern "C" int tgis_gptq_dequant_f16(const void* prepared, void* w_out, int64_t K, int64_t N, int64_t groups,
                                     void* stream) {
    TGIS_CHECK_ARG(prepared && w_out && K > 0 && N > 0 && groups > 0 && K % groups == 0,
                   "tgis_gptq_dequant_f16: bad arguments");
    PrepLayout p = prep_layout(K, N, groups);
    int64_t total = p.NT * p.KS * 256;
    hipLaunchKernelGGL(gptq_dequant_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)prepared, p.offB, p.offC, (f16*)w_out, (int)K, (int)N, (int)groups,
                       (int)(K / groups), (int)p.NT, (int)p.KS);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}
