// Round 4: the int4 decode GEMM re-cut from its byte budget instead of from the round-1 kernel.
//   - a wave owns CT 32-column tiles (not one) over its own k-range and takes the activation straight from L2 into the MFMA
//     A operand (16 bytes per lane per MFMA, 64 contiguous bytes per lane per k64-step, shared by the CT tiles): no LDS
//     staging, no chunk hand-offs, no barrier until the k-part sum at the very end;
//   - WK waves of a block split k; their fp32 sums meet once in LDS (distributed finish, fixed order);
//   - DEPTH k64-steps of weights + scales + activation in flight per wave, refilled in place.
// Same prepared image as libtgis_hip.so (tgis_gptq_prepare): wq[NT][KS][64 lanes][4] u32, sz[NT][G][32] {scale, 1024+z+1}.
// Self-contained: builds random images, checks sampled columns against a CPU sum, times cold weights (rotating sets).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/floor/wide tools/floor/wide.hip && tools/floor/wide
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
#define PIN_SGPR(p) asm volatile("" : "+s"(p))

__device__ __forceinline__ uint32_t and_or(uint32_t q, uint32_t mask, uint32_t ex) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(q), "s"(mask), "v"(ex));
    return r;
}
__device__ __forceinline__ f16x8 dequant8(uint32_t q, f16x2 zc, f16x2 zd, f16x2 sc, uint32_t EX, uint32_t M0, uint32_t M1) {
    const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
    uint32_t q2 = q >> 8;
    uint32_t a0 = and_or(q, M0, EX), a1 = and_or(q, M1, EX), a2 = and_or(q2, M0, EX), a3 = and_or(q2, M1, EX);
    f16x2 h0 = (__builtin_bit_cast(f16x2, a0) - zc) * sc;
    f16x2 h1 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a1), r16, zd) * sc;
    f16x2 h2 = (__builtin_bit_cast(f16x2, a2) - zc) * sc;
    f16x2 h3 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a3), r16, zd) * sc;
    u32x4 p = {__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1), __builtin_bit_cast(uint32_t, h2),
               __builtin_bit_cast(uint32_t, h3)};
    return __builtin_bit_cast(f16x8, p);
}

struct Args {
    const f16* x; int ldx;
    const f16* xf;         // the same activation in fragment order
    const uint8_t* prep; int64_t offB;
    f16* out; int ldo;
    float* slabs;          // [S][32][NT*32]
    int M, K, N, NT, KS, G, spg_shift;
    int S, steps;          // global splits, real k64-steps (K / 64)
    int lay;               // wide2_gemm, timing only (results are wrong): 0 = the prepared image [tile][step]; 1 = step-major
                           // [local step][tile][k part] (what all waves read at one instant is one dense run); 2 = [local step][k part][tile]
    long long* trace;      // [blocks][WK][8] s_memtime stamps (nullptr: off)
    // EPI 3.. (rope epilogue as in gptq_wide_body.h ACT 3)
    const int32_t* positions; const int32_t* slots; const f16* cosb; const f16* sinb; f16* kpool; f16* vpool;
    int rH, rHkv, rD;
    // EPI 10 (split-K finished in the launch: the last split of a column group sums the slabs, adds the residual, writes
    // h in row-major and fragment order and the rows' partial sums of squares); EPI 11 (the consumer of that: A = h * w,
    // rstd from the partial sums in the epilogue)
    unsigned* counters; const f16* resid; f16* hout; f16* hfrag; float* ssq;   // ssq [32][NT]
    const f16* normw; const float* ssq_in; int ssq_n; float eps;
    // EPI 12 (norm-ahead blocks: the add + RMSNorm in front of the GEMM runs on the launch's first 32 workgroups, the GEMM
    // workgroups request their weights and wait for a counter before they ask for x) and its two-launch baseline
    const float* nslabs; const f16* nres; f16* nres_out; f16* nxf; unsigned* nflag; unsigned ntarget;
};

// the add + RMSNorm of norm.hip for one row of 4096 by 512 threads: 4 fp32 slabs + residual -> residual stream, normed row in
// fragment order.  SC1: write-through stores, drained, then the arrival counter (the row is read by other workgroups of the launch)
template <bool SC1>
__device__ __forceinline__ void norm_row_emul(const Args& a, int row, float* sh) {
    const int c = threadIdx.x;  // chunk of 8 elements
    const int H = 4096;
    f32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        const float* p = a.nslabs + ((int64_t)s2 * 32 + row) * H + c * 8;
        lo += *reinterpret_cast<const f32x4*>(p);
        hi += *reinterpret_cast<const f32x4*>(p + 4);
    }
    const f16x8 r = *reinterpret_cast<const f16x8*>(a.nres + (int64_t)row * H + c * 8);
    const f16x8 w = *reinterpret_cast<const f16x8*>(a.normw + c * 8);
    float v[8], ss = 0.f;
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        v[e] = (float)(f16)(e < 4 ? lo[e] : hi[e - 4]) + (float)r[e];
        o[e] = (f16)v[e];
        ss += v[e] * v[e];
    }
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(a.nres_out, 0, 0x7FFFFFFF, 0x00020000);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(a.nxf, 0, 0x7FFFFFFF, 0x00020000);
    if (SC1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rr, (uint32_t)(((int64_t)row * H + c * 8) * 2), 0, 16);
    else *reinterpret_cast<f16x8*>(a.nres_out + (int64_t)row * H + c * 8) = o;
#pragma unroll
    for (int of = 32; of > 0; of >>= 1) ss += __shfl_xor(ss, of, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int k = 0; k < 8; ++k) tot += sh[k];
    const float rstd = rsqrtf(tot / H + a.eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)(v[e] * rstd * (float)w[e]);
    const int k = c * 8;
    const int64_t off = ((((k >> 6) << 2) + ((k >> 3) & 3)) * 64 + ((k >> 5) & 1) * 32 + (row & 31)) * 8;
    if (SC1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rx, (uint32_t)(off * 2), 0, 16);
    else *reinterpret_cast<f16x8*>(a.nxf + off) = o;
}
__global__ __launch_bounds__(512) void norm_emul_kernel(Args a) {
    __shared__ float sh[16];
    norm_row_emul<false>(a, blockIdx.x, sh);
}
__device__ __forceinline__ int64_t k_off(int tok, int d, int D) {
    return ((int64_t)(((tok >> 4) * (D >> 3) + (d >> 3)) * 16 + (tok & 15)) << 3) + (d & 7);
}
__device__ __forceinline__ int v_col(int tok) { int i = tok & 15; return (i >> 2) * 8 + (tok >> 4) * 4 + (i & 3); }
#define STAMP(i) do { if (TR) { stamp[i] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } } while (0)

// MODE: 0 full; 1 no arithmetic (loads, waits, reduce, stores); 2 no x loads (A = weights); 3 no weight loads beyond prologue
template <int CT, int WK, int DEPTH, int MODE, int XF = 0, int ORD = 0, bool TR = false, int EPI = 0>
__global__ __launch_bounds__(64 * WK) void wide_gemm(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (EPI == 12 && blockIdx.x < 32) {   // the norm-ahead workgroups
        norm_row_emul<true>(a, blockIdx.x, reinterpret_cast<float*>(smem));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.nflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int cg = EPI == 12 ? blockIdx.x - 32 : blockIdx.x, split = blockIdx.y;
    // this split's steps, then this wave's share of them (contiguous)
    const int sp_len = (a.steps + a.S - 1) / a.S;
    const int sb = split * sp_len, se = min(a.steps, sb + sp_len);
    const int len = max(se - sb, 0);
    const int s0 = sb + (len * wk) / WK, s1 = sb + (len * (wk + 1)) / WK;
    const int mrows = min(32, a.M);
    long long stamp[8];
    STAMP(0);
    if (TR) stamp[7] = __builtin_amdgcn_s_memrealtime();

    const char* wt[CT];
    const char* st[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int nt = min(cg * CT + t, a.NT - 1);
        wt[t] = reinterpret_cast<const char*>(a.prep) + (int64_t)nt * a.KS * 1024;
        st[t] = reinterpret_cast<const char*>(a.prep + a.offB) + (int64_t)nt * a.G * 128;
    }
    const uint32_t woff = lane * 16, szoff = (lane & 31) * 4;
    const int sclamp = max(s1 - 1, s0);
    const uint32_t xlane = (uint32_t)(min(lane & 31, mrows - 1) * a.ldx * 2 + (lane >> 5) * 64);
    const char* xb = reinterpret_cast<const char*>(a.x);

    u32x4 wq[DEPTH][CT];
    uint32_t sz[DEPTH][CT];
    f16x8 xa[DEPTH][4];
    f16x8 wn[EPI == 11 ? DEPTH : 1][4];
    auto load_x = [&](int d, int step) {
        if (XF) {   // fragment-major activation: [step][i][lane][8 halves], 1 KiB per load
            // (MODE 8: every wave re-reads step 0 — the same four KiB, resident in the CU's L1: the instructions without the L2 traffic;
            //  MODE 9: only the first of the four loads is real)
            const char* p = reinterpret_cast<const char*>(a.xf) + (MODE == 8 ? (int64_t)0 : (int64_t)min(step, sclamp) * 4096);
            PIN_SGPR(p);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 9 && i > 0) { xa[d][i] = xa[d][0]; continue; }
                xa[d][i] = *(const GLOBAL_AS f16x8*)(p + woff + i * 1024);
            }
            if (EPI == 11) {   // the norm weight of the same k slots: two distinct 16-byte pieces per load
                const char* q = reinterpret_cast<const char*>(a.normw) + (int64_t)min(step, sclamp) * 128;
                PIN_SGPR(q);
#pragma unroll
                for (int i = 0; i < 4; ++i) wn[d][i] = *(const GLOBAL_AS f16x8*)(q + (lane >> 5) * 16 + i * 32);
            }
            return;
        }
        const char* p = xb + (int64_t)min(step, sclamp) * 128;
        PIN_SGPR(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) xa[d][i] = *(const GLOBAL_AS f16x8*)(p + xlane + i * 16);
    };
    auto load_sz = [&](int d, int step) {
        const int g = min(min(step, sclamp) >> a.spg_shift, a.G - 1);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const char* p = st[t] + (int64_t)g * 128;
            PIN_SGPR(p);
            const uint32_t v = *(const GLOBAL_AS uint32_t*)(p + szoff);
            sz[d][t] = step < s1 ? v : 0u;
        }
    };
    auto load_w = [&](int d, int step) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const char* p = wt[t] + (int64_t)min(step, sclamp) * 1024;
            PIN_SGPR(p);
            wq[d][t] = __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(p + woff));
        }
    };

    constexpr int NR_ = 16 / WK;
    int32_t rpos[NR_], rslot[NR_];
    if (EPI >= 3) {
#pragma unroll
        for (int j = 0; j < NR_; ++j) {
            const int r = wk * NR_ + j;
            const int m = min((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), mrows - 1);
            rpos[j] = a.positions[m];
            rslot[j] = a.slots[m];
        }
    }
    float ssq_part[NR_][2][2];
    if (EPI == 11) {
#pragma unroll
        for (int j = 0; j < NR_; ++j)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int r = wk * NR_ + j;
                const int m = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float* pp = a.ssq_in + (int64_t)m * a.ssq_n;   // ssq_n <= 128: two per lane
                ssq_part[j][hh][0] = lane < a.ssq_n ? pp[lane] : 0.f;
                ssq_part[j][hh][1] = lane + 64 < a.ssq_n ? pp[lane + 64] : 0.f;
            }
    }
    uint32_t touch = 0;
    if (EPI == 7) {   // touch the cache lines the epilogue will write (k / v heads), so that its partial-line stores hit
        const int per = a.rD >> 5, c = lane & 31;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int nt = min(cg * CT + t, a.NT - 1);
            const int head = nt / per, tt = nt - head * per;
            const bool roth = head < a.rH + a.rHkv;
            const int d = roth ? ((c < 16) ? 16 * tt + c : (a.rD >> 1) + 16 * tt + (c - 16)) : 32 * tt + c;
            if (head >= a.rH) {
#pragma unroll
                for (int j = 0; j < NR_; ++j) {
                    const int page = rslot[j] >> 5, tok = rslot[j] & 31;
                    const f16* p = roth ? a.kpool + ((int64_t)page * a.rHkv + (head - a.rH)) * 32 * a.rD + k_off(tok, d, a.rD)
                                        : a.vpool + ((int64_t)page * a.rHkv + (head - a.rH - a.rHkv)) * 32 * a.rD + (int64_t)d * 32 + v_col(tok);
                    touch += *(const volatile uint16_t*)p;
                }
            }
        }
    }
    uint32_t EXr = 0x64006400u, M0r = 0x000F000Fu, M1r = 0x00F000F0u;
    asm volatile("" : "+v"(EXr));
    asm volatile("" : "+s"(M0r), "+s"(M1r));
    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // prologue.  ORD 0: type-major (all x, all scales, all weights); ORD 1: step-major (what step 0 needs first)
    if (EPI == 12) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { load_sz(d, s0 + d); load_w(d, s0 + d); }
        for (unsigned spins = 0; (int)(__builtin_amdgcn_readfirstlane(__hip_atomic_load(a.nflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - a.ntarget) < 0; ++spins) {
            __builtin_amdgcn_s_sleep(2);
            if (spins > (1u << 20)) break;   // never hang the device
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load_x(d, s0 + d);
    } else if (ORD == 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if (MODE != 2 && (MODE < 4 || MODE > 5)) load_x(d, s0 + d);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if (MODE != 4) load_sz(d, s0 + d);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load_w(d, s0 + d);
    } else {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (MODE != 4) load_sz(d, s0 + d);
            load_w(d, s0 + d);
            if (MODE != 2 && (MODE < 4 || MODE > 5)) load_x(d, s0 + d);
        }
    }

    auto consume = [&](int d) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const f16x2 szh = __builtin_bit_cast(f16x2, sz[d][t]);
            const f16 zc1 = szh[1];
            const f16 zd1 = (f16)960.f - zc1;
            const f16x2 zc = {zc1, zc1}, zd = {zd1, zd1}, sc = {szh[0], szh[0]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 6) {
                    const u32x4 raw = {wq[d][t][i], wq[d][t][i] ^ EXr, sz[d][t], wq[d][t][i]};
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[d][i], __builtin_bit_cast(f16x8, raw), acc[t], 0, 0, 0);
                } else if (MODE == 7) {
                    const f16x8 b = dequant8(wq[d][t][i], zc, zd, sc, EXr, M0r, M1r);
                    const f16x8 pr = b * xa[d][i];
                    acc[t][i] += (float)pr[0] + (float)pr[2] + (float)pr[4] + (float)pr[6];
                } else if (MODE == 4) {
                    acc[t][i] += __builtin_bit_cast(float, wq[d][t][i]);
                } else if (MODE == 5) {
                    acc[t][i] += __builtin_bit_cast(float, wq[d][t][i]) + (float)szh[0];
                } else if (MODE == 1) {
                    acc[t][i] += __builtin_bit_cast(float, wq[d][t][i]) + (float)xa[d][i][0] + (float)szh[0];
                } else {
                    const f16x8 b = dequant8(wq[d][t][i], zc, zd, sc, EXr, M0r, M1r);
                    if (EPI == 11 && t == 0) xa[d][i] = xa[d][i] * wn[d][i];
                    const f16x8 av = MODE == 2 ? __builtin_bit_cast(f16x8, wq[d][(t + 1) % CT]) : xa[d][i];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b, acc[t], 0, 0, 0);
                }
            }
        }
    };

    f16 rcos9[CT][16 / WK], rsin9[CT][16 / WK];
    if (EPI == 9) {
        const int per = a.rD >> 5;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int nt = min(cg * CT + t, a.NT - 1);
            const int tt = nt - (nt / per) * per;
            const int dr = 16 * tt + (lane & 15);
            const bool roth = nt / per < a.rH + a.rHkv;
#pragma unroll
            for (int j = 0; j < 16 / WK; ++j) {
                rcos9[t][j] = roth ? a.cosb[(int64_t)rpos[j] * (a.rD >> 1) + dr] : (f16)1.f;
                rsin9[t][j] = roth ? a.sinb[(int64_t)rpos[j] * (a.rD >> 1) + dr] : (f16)0.f;
            }
        }
    }
    STAMP(1);
    int s = s0;
    for (; s + DEPTH < s1; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            consume(d);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE != 4) load_sz(d, s + d + DEPTH);
            if (MODE != 3) load_w(d, s + d + DEPTH);
            if (MODE != 2 && (MODE < 4 || MODE > 5)) load_x(d, s + d + DEPTH);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    STAMP(2);
    // the last group: only the steps that exist (wave-uniform branches; nothing is loaded any more)
    if (s < s1) consume(0);
    STAMP(3);
#pragma unroll
    for (int d = 1; d < DEPTH; ++d)
        if (s + d < s1) consume(d);
    STAMP(4);

    constexpr int NR = 16 / WK;
    f16 rcos[CT][NR], rsin[CT][NR];
    if (EPI == 9) {
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int j = 0; j < NR; ++j) { rcos[t][j] = rcos9[t][j]; rsin[t][j] = rsin9[t][j]; }
    } else if (EPI >= 3 && EPI != 5) {
        const int per = a.rD >> 5;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int nt = min(cg * CT + t, a.NT - 1);
            const int tt = nt - (nt / per) * per;
            const int dr = 16 * tt + (lane & 15);
            const bool roth = nt / per < a.rH + a.rHkv;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                rcos[t][j] = roth ? a.cosb[(int64_t)rpos[j] * (a.rD >> 1) + dr] : (f16)1.f;
                rsin[t][j] = roth ? a.sinb[(int64_t)rpos[j] * (a.rD >> 1) + dr] : (f16)0.f;
            }
        }
    }
    // ---- k-part sum through LDS, distributed finish: wave wk ends up with registers [wk NR, (wk+1) NR) of every tile ----
    float* red = reinterpret_cast<float*>(smem);  // [WK][CT][16 registers][64 lanes]: every access is 64 consecutive words
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        float* dst = red + ((wk * CT + t) << 10) + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r << 6] = acc[t][r];
    }
    __syncthreads();
    STAMP(5);
    float fin[CT][NR];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
#pragma unroll
        for (int k2 = 0; k2 < WK; ++k2) {
            const float* src = red + ((k2 * CT + t) << 10) + ((wk * NR) << 6) + lane;
#pragma unroll
            for (int j = 0; j < NR; ++j) fin[t][j] = k2 == 0 ? src[j << 6] : fin[t][j] + src[j << 6];
        }
    }
    const int c = lane & 31;
    float rstd[NR];
    if (EPI == 11) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            float v0 = ssq_part[j][0][0] + ssq_part[j][0][1], v1 = ssq_part[j][1][0] + ssq_part[j][1][1];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_xor(v0, o, 64); v1 += __shfl_xor(v1, o, 64); }
            rstd[j] = rsqrtf(((lane >> 5) ? v1 : v0) / (float)a.K + a.eps);
        }
    }
    if (EPI == 10 && a.S > 1) {
        // ---- split-K finished in the launch ----
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, 0x7FFFFFFF, 0x00020000);
        const int ld = a.NT * 32;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int nt = min(cg * CT + t, a.NT - 1);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = wk * NR + j;
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, fin[t][j]), rs,
                                                      (uint32_t)((((int64_t)split * 32 + m) * ld + nt * 32 + c) * 4), 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* lastp = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.counters + cg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old + 1u == (unsigned)a.S;
            if (last) __hip_atomic_store(a.counters + cg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *lastp = last;
        }
        __syncthreads();
        if (!*lastp) return;
        float part[CT][NR][4];   // S <= 4 here
        f16 res[CT][NR];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int nt = min(cg * CT + t, a.NT - 1);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = wk * NR + j;
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
                    part[t][j][s2] = s2 < a.S ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs,
                                         (uint32_t)((((int64_t)s2 * 32 + m) * ld + nt * 32 + c) * 4), 0, 16)) : 0.f;
                res[t][j] = a.resid[(int64_t)min(m, mrows - 1) * a.N + nt * 32 + c];
            }
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int nt = cg * CT + t;
            if (nt >= a.NT) break;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = wk * NR + j;
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float o = (float)(f16)(((part[t][j][0] + part[t][j][1]) + part[t][j][2]) + part[t][j][3]);
                const float h = o + (float)res[t][j];
                const f16 hq = (f16)h;
                float sq = h * h;
#pragma unroll
                for (int of = 16; of > 0; of >>= 1) sq += __shfl_xor(sq, of, 64);
                if (m < mrows) {
                    a.hout[(int64_t)m * a.N + nt * 32 + c] = hq;
                    const int k = nt * 32 + c;
                    a.hfrag[((((k >> 6) << 2) + ((k >> 3) & 3)) * 64 + ((k >> 5) & 1) * 32 + (m & 31)) * 8 + (k & 7)] = hq;
                    if (c == 0) a.ssq[(int64_t)m * a.NT + nt] = sq;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int nt = cg * CT + t;
        if (nt >= a.NT) break;
        const int n = nt * 32 + c;
        if (EPI >= 3 && EPI < 10) {
            const int per = a.rD >> 5;
            const int head = nt / per, tt = nt - head * per;
            const bool roth = head < a.rH + a.rHkv;
            const int d = roth ? ((c < 16) ? 16 * tt + c : (a.rD >> 1) + 16 * tt + (c - 16)) : 32 * tt + c;
            const int col = head * a.rD + d;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = wk * NR + j;
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float mine = (float)(f16)(fin[t][j]);
                float o = mine;
                if (roth && EPI != 5) {
                    const float other = __shfl_xor(mine, 16, 64);
                    const float cf = (float)rcos[t][j], sf = (float)rsin[t][j];
                    o = (c < 16) ? mine * cf - other * sf : other * sf + mine * cf;
                }
                const f16 oh = (f16)o;
                if (m < mrows) {
                    if (head < a.rH || EPI == 6) {
                        a.out[(int64_t)m * a.ldo + col] = oh;
                    } else {
                        const int page = rslot[j] >> 5, tok = rslot[j] & 31;
                        if (roth)
                            a.kpool[((int64_t)page * a.rHkv + (head - a.rH)) * 32 * a.rD + k_off(tok, d, a.rD)] = oh;
                        else if (EPI == 8 || EPI == 9)   // V page as [tok / 8][D][8]: a token's d run is 16-byte strided (16 lines per head instead of 64)
                            a.vpool[((int64_t)page * a.rHkv + (head - a.rH - a.rHkv)) * 32 * a.rD + (int64_t)(tok >> 3) * 8 * a.rD + d * 8 + (tok & 7)] = oh;
                        else if (EPI == 4)   // v written row-major into the q tensor's space: what does the scatter cost?
                            a.out[(int64_t)m * a.ldo + col] = oh;
                        else
                            a.vpool[((int64_t)page * a.rHkv + (head - a.rH - a.rHkv)) * 32 * a.rD + (int64_t)d * 32 + v_col(tok)] = oh;
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = wk * NR + j;
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (a.S == 1) {
                if (m < mrows && n < a.N) a.out[(int64_t)m * a.ldo + n] = (f16)(EPI == 11 ? fin[t][j] * rstd[j] : fin[t][j]);
            } else {
                a.slabs[((int64_t)split * 32 + m) * (a.NT * 32) + n] = fin[t][j];
            }
        }
    }
    STAMP(6);
    if (EPI == 7 && touch == 0x12345u) a.out[0] = (f16)1.f;
    if (TR && lane == 0) {
        long long* tp = a.trace + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * WK + wk) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) tp[i] = stamp[i];
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Round 5: the same unit re-cut to answer VERDICT r04 item 1 (ramp and drain of a launch):
//   DW / DX  k64-steps of weights (+ scales) / of activation in flight per wave, separately: weights come from HBM (long
//            latency, 1 KiB per tile-step and 4 VGPRs), the activation from L2 (short latency, 4 KiB per step, 16 VGPRs);
//   ORD 2    prologue requests every weight step before any activation;
//   PRIO     s_setprio 1 for the younger half of the block's waves (a CU serves its older waves first);
//   PRE      every cache line of the kernel arguments is requested at entry (one scalar round trip instead of three);
//   WK 4     four-wave blocks: 32 KiB of LDS at CT 2, <= 128 VGPRs -> up to four blocks per CU, more blocks than CUs.
//   KB       k64-steps that every wave of the older half of the block takes over from its partner in the younger half: the
//            younger half starts its stream ~2.4 k ticks later (its prologue requests queue behind the older half's in the
//            CU's 64 B/clk address path) and everybody waits for it at the k-part exchange (profiles/r05_wide_timeline.log);
//   PB       a block barrier between the prologue's step-0 requests and its later steps (every wave's first step ahead of
//            anybody's second);
//   EPI      0 plain; 20 SiLU * up on the interleaved image, row-major; 21 the same in fragment order (2-byte stores, the
//            product's OUTF); 22 fragment order through LDS: 16-byte stores.
//   AR 1     "lean" arithmetic for group size 128 (two k64-steps per group, wave ranges aligned to groups): the B operand is the
//            raw nibble as 1024 + q (low positions) / 64 + q (high positions): 5 VALU per 8 weights instead of 13; offsets and
//            zero point leave through ONE more MFMA per tile and group whose A operand carries the lane's partial row sums of
//            x over the two position classes (hi + lo f16 parts, v_dot2 from the fragments already in registers) and whose B
//            operand is {-zc, -zc, zd, zd}; the group's scale is applied to the group's fp32 sum (16 fma per tile and group).
template <int CT, int WK, int DW, int DX, int ORD, int PRIO, int PRE, bool TR, int KB = 0, int PB = 0, int EPI = 0, int AR = 0>
__global__ __launch_bounds__(64 * WK) void wide2_gemm(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(DW % DX == 0, "the activation ring divides the weight ring");
    if (PRE) {
        const int p0 = a.rD; const void* p1 = a.ssq; const void* p2 = a.nflag; const void* p3 = a.slabs;
        asm volatile("" :: "s"(p0), "s"(p1), "s"(p2), "s"(p3));
    }
    const int lane = threadIdx.x & 63;
    const int wk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (PRIO == 1) { if (wk >= WK / 2) __builtin_amdgcn_s_setprio(1); }
    if (PRIO == 2) { if (wk < WK / 2) __builtin_amdgcn_s_setprio(1); }
    const int cg = blockIdx.x, split = blockIdx.y;
    const int sp_len = (a.steps + a.S - 1) / a.S;
    const int sb = split * sp_len, se = min(a.steps, sb + sp_len);
    const int len = max(se - sb, 0);
    // boundaries f(w) = len w / WK + kb (min(w, WK/2) - max(w - WK/2, 0)): the older half gets kb steps more per wave
    const int kb = min(KB, max(len / WK - 1, 0));
    auto bound = [&](int w) { return sb + (len * w) / WK + kb * (min(w, WK / 2) - max(w - WK / 2, 0)); };
    const int s0 = bound(wk), s1 = bound(wk + 1);
    const int mrows = min(32, a.M);
    long long stamp[8];
    STAMP(0);
    if (TR) stamp[7] = __builtin_amdgcn_s_memrealtime();

    const char* wt[CT];
    const char* st[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int nt = min(cg * CT + t, a.NT - 1);
        wt[t] = reinterpret_cast<const char*>(a.prep) + (int64_t)nt * a.KS * 1024;
        st[t] = reinterpret_cast<const char*>(a.prep + a.offB) + (int64_t)nt * a.G * 128;
    }
    const uint32_t woff = lane * 16, szoff = (lane & 31) * 4;
    const int sclamp = max(s1 - 1, s0);

    u32x4 wq[DW][CT];
    uint32_t sz[DW][CT];
    f16x8 xa[DX][4];
    auto load_x = [&](int d, int step) {
        const char* p = reinterpret_cast<const char*>(a.xf) + (int64_t)min(step, sclamp) * 4096;
        PIN_SGPR(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) xa[d][i] = *(const GLOBAL_AS f16x8*)(p + woff + i * 1024);
    };
    auto load_w = [&](int d, int step) {
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
            if (a.lay) {
                const int64_t nt = min(cg * CT + t, a.NT - 1), parts = (int64_t)WK * a.S, part = split * WK + wk;
                const int64_t idx = a.lay == 1 ? ((int64_t)(sc - s0) * a.NT + nt) * parts + part : ((int64_t)(sc - s0) * parts + part) * a.NT + nt;
                p = reinterpret_cast<const char*>(a.prep) + min(idx, (int64_t)a.NT * a.KS - 1) * 1024;
            }
            PIN_SGPR(p);
            wq[d][t] = __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(p + woff));
        }
    };
    uint32_t EXr = 0x64006400u, M0r = 0x000F000Fu, M1r = 0x00F000F0u;
    asm volatile("" : "+v"(EXr));
    asm volatile("" : "+s"(M0r), "+s"(M1r));
    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    if (ORD == 2) {
#pragma unroll
        for (int d = 0; d < DW; ++d) load_w(d, s0 + d);
#pragma unroll
        for (int d = 0; d < DX; ++d) load_x(d, s0 + d);
    } else {
#pragma unroll
        for (int d = 0; d < DW; ++d) {
            load_w(d, s0 + d);
            if (d < DX) load_x(d, s0 + d);
            if (PB && d == 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    auto consume = [&](int d) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const f16x2 szh = __builtin_bit_cast(f16x2, sz[d][t]);
            const f16 zc1 = szh[1];
            const f16 zd1 = (f16)960.f - zc1;
            const f16x2 zc = {zc1, zc1}, zd = {zd1, zd1}, sc = {szh[0], szh[0]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 b = dequant8(wq[d][t][i], zc, zd, sc, EXr, M0r, M1r);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[d % DX][i], b, acc[t], 0, 0, 0);
            }
        }
    };
    f32x16 accg[AR ? CT : 1];
    float SL = 0.f, SH = 0.f;
    uint32_t EXH = 0x54005400u;
    asm volatile("" : "+v"(EXH));
    auto consume_lean = [&](int d) {
        static_assert(!AR || (DW == 2 && DX == 2), "lean arithmetic: the ring is one group");
        const f16x2 one = {(f16)1.f, (f16)1.f};
        if (d == 0) { SL = 0.f; SH = 0.f; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (element pairs, NOT __builtin_bit_cast(f16x2, <element of a u32x4>): hipcc 7.2 folds the four of them onto dword 0)
            const f16x8 xv = xa[d][i];
            SL = __builtin_amdgcn_fdot2(f16x2{xv[0], xv[1]}, one, SL, false);
            SH = __builtin_amdgcn_fdot2(f16x2{xv[2], xv[3]}, one, SH, false);
            SL = __builtin_amdgcn_fdot2(f16x2{xv[4], xv[5]}, one, SL, false);
            SH = __builtin_amdgcn_fdot2(f16x2{xv[6], xv[7]}, one, SH, false);
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t q = wq[d][t][i], q2 = q >> 8;
                // plain C, not the asm and_or: hipcc does not pad an asm VALU result that an MFMA reads next (NOTEBOOK.md section 6,
                // round 3: stale rows); the same expression in C compiles to v_and_or_b32 with the hazard handled
                const u32x4 raw = {(q & M0r) | EXr, (q & M1r) | EXH, (q2 & M0r) | EXr, (q2 & M1r) | EXH};
                if (d == 0 && i == 0)
                    accg[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[d][i], __builtin_bit_cast(f16x8, raw),
                                                                     f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
                else
                    accg[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[d][i], __builtin_bit_cast(f16x8, raw), accg[t], 0, 0, 0);
            }
        }
        if (d == 1) {
            const f16 slh = (f16)SL, shh = (f16)SH;
            const f16 sll = (f16)(SL - (float)slh), shl = (f16)(SH - (float)shh);
            const f16x2 a0 = {slh, sll}, a1 = {shh, shl};
            const u32x4 ae = {__builtin_bit_cast(uint32_t, a0), __builtin_bit_cast(uint32_t, a1), 0u, 0u};
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const f16x2 szh = __builtin_bit_cast(f16x2, sz[d][t]);
                const f16 nzc = -szh[1], zd1 = (f16)960.f - szh[1];
                const f16x2 b0 = {nzc, nzc}, b1 = {zd1, zd1};
                const u32x4 be = {__builtin_bit_cast(uint32_t, b0), __builtin_bit_cast(uint32_t, b1), 0u, 0u};
                accg[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ae), __builtin_bit_cast(f16x8, be), accg[t], 0, 0, 0);
                const float sf = (float)szh[0];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = __builtin_fmaf(sf, accg[t][r], acc[t][r]);
            }
        }
    };
    STAMP(1);
    int s = s0;
    for (; s + DW < s1; s += DW) {
#pragma unroll
        for (int d = 0; d < DW; ++d) {
            if (AR) consume_lean(d); else
            consume(d);
            __builtin_amdgcn_sched_barrier(0);
            load_w(d, s + d + DW);
            load_x(d % DX, s + d + DX);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    STAMP(2);
    // the last group: the steps that exist; the activation ring is shorter than the weight ring, so it is still refilled
#pragma unroll
    for (int d = 0; d < DW; ++d) {
        if (s + d < s1) { if (AR) consume_lean(d); else consume(d); }
        if (d + DX < DW) {
            __builtin_amdgcn_sched_barrier(0);
            load_x(d % DX, s + d + DX);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (d == 0) STAMP(3);
    }
    STAMP(4);
    constexpr int NR = 16 / WK;
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        float* dst = red + ((wk * CT + t) << 10) + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r << 6] = acc[t][r];
    }
    __syncthreads();
    STAMP(5);
    float fin[CT][NR];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
#pragma unroll
        for (int k2 = 0; k2 < WK; ++k2) {
            const float* src = red + ((k2 * CT + t) << 10) + ((wk * NR) << 6) + lane;
#pragma unroll
            for (int j = 0; j < NR; ++j) fin[t][j] = k2 == 0 ? src[j << 6] : fin[t][j] + src[j << 6];
        }
    }
    const int c = lane & 31;
    if (EPI >= 20) {
        // SiLU * up on the interleaved image: lanes c < 16 hold gate column j2, lanes c + 16 the matching up column
        const int half = a.N >> 1;
        f16* stg = reinterpret_cast<f16*>(smem + (size_t)WK * CT * 4096);   // [32 rows][CT * 16] halves (EPI 22)
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int nt = cg * CT + t;
            const int j2 = nt * 16 + (c & 15);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = wk * NR + j;
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float mine = (float)(f16)fin[t][j];
                const float other = __shfl_xor(mine, 16, 64);
                const float sl = mine / (1.f + __expf(-mine));
                const f16 o = (f16)((float)(f16)sl * other);
                if (EPI == 22) {
                    if (c < 16) stg[m * (CT * 16) + t * 16 + c] = o;
                } else if (c < 16 && nt < a.NT && j2 < half && m < mrows) {
                    if (EPI == 21) {
                        const int k = j2;
                        a.out[((((k >> 6) << 2) + ((k >> 3) & 3)) * 64 + ((k >> 5) & 1) * 32 + (m & 31)) * 8 + (k & 7)] = o;
                    } else {
                        a.out[(int64_t)m * half + j2] = o;
                    }
                }
            }
        }
        if (EPI == 22) {
            __syncthreads();
            constexpr int CH = CT * 2;   // 16-byte chunks per row
            for (int id = threadIdx.x; id < 32 * CH; id += 64 * WK) {
                const int m = id / CH, ch = id - m * CH;
                const int k = cg * CT * 16 + ch * 8;
                if (k < half && m < mrows) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(stg + m * (CT * 16) + ch * 8);
                    *reinterpret_cast<u32x4*>(a.out + ((((k >> 6) << 2) + ((k >> 3) & 3)) * 64 + ((k >> 5) & 1) * 32 + m) * 8) = v;
                }
            }
        }
    } else {
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int nt = cg * CT + t;
        if (nt >= a.NT) break;
        const int n = nt * 32 + c;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = wk * NR + j;
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (a.S == 1) {
                if (m < mrows && n < a.N) a.out[(int64_t)m * a.ldo + n] = (f16)fin[t][j];
            } else {
                a.slabs[((int64_t)split * 32 + m) * (a.NT * 32) + n] = fin[t][j];
            }
        }
    }
    }
    STAMP(6);
    if (TR && lane == 0) {
        long long* tp = a.trace + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * WK + wk) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) tp[i] = stamp[i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
struct Image {
    int K, N, NT, KS, G, gs;
    int64_t offB, total;
    std::vector<uint8_t> host;
};
static uint64_t rng_state = 0x1234567ull;
static inline uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }

static Image make_image(int K, int N, int gs) {
    Image im;
    im.K = K; im.N = N; im.gs = gs; im.G = (K + gs - 1) / gs;
    im.NT = (N + 31) / 32; im.KS = ((K + 255) / 256) * 4 + 1;
    im.offB = (int64_t)im.NT * im.KS * 1024;
    im.total = (im.offB + (int64_t)im.NT * im.G * 128 + 255) & ~255ll;
    im.host.resize(im.total);
    uint32_t* w = reinterpret_cast<uint32_t*>(im.host.data());
    for (int64_t i = 0; i < im.offB / 4; ++i) w[i] = rnd() ^ (rnd() << 16);
    // rows >= K hold zero nibbles in the real image; K is a multiple of 64 here so only whole steps are padding
    for (int nt = 0; nt < im.NT; ++nt)
        for (int ks = K / 64; ks < im.KS; ++ks) std::fill_n(w + ((int64_t)nt * im.KS + ks) * 256, 256, 0u);
    uint32_t* sz = reinterpret_cast<uint32_t*>(im.host.data() + im.offB);
    for (int64_t i = 0; i < (int64_t)im.NT * im.G * 32; ++i) {
        f16 s = (f16)(0.005f + 0.01f * (rnd() % 1000) / 1000.f);
        f16 z = (f16)(1024.f + (rnd() % 14) + 1);
        uint16_t sb, zb;
        memcpy(&sb, &s, 2); memcpy(&zb, &z, 2);
        sz[i] = sb | ((uint32_t)zb << 16);
    }
    return im;
}
static double ref_dot(const Image& im, const std::vector<f16>& x, int ldx, int m, int n) {
    static const int order[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    const uint32_t* w = reinterpret_cast<const uint32_t*>(im.host.data());
    const uint32_t* sz = reinterpret_cast<const uint32_t*>(im.host.data() + im.offB);
    const int nt = n / 32, c = n % 32;
    double s = 0;
    for (int ks = 0; ks < im.K / 64; ++ks)
        for (int h = 0; h < 2; ++h)
            for (int i = 0; i < 4; ++i) {
                const uint32_t q = w[(((int64_t)nt * im.KS + ks) * 64 + h * 32 + c) * 4 + i];
                for (int j = 0; j < 8; ++j) {
                    const int k = ks * 64 + h * 32 + i * 8 + order[j];
                    const uint32_t e = sz[((int64_t)nt * im.G + k / im.gs) * 32 + c];
                    uint16_t sb = e & 0xffff, zb = e >> 16;
                    f16 sc, zc;
                    memcpy(&sc, &sb, 2); memcpy(&zc, &zb, 2);
                    const f16 wv = (f16)(((float)((q >> (4 * j)) & 15) + 1024.f - (float)zc) * (float)sc);
                    s += (double)(float)wv * (double)(float)x[(int64_t)m * ldx + k];
                }
            }
    return s;
}

static const f16* g_xf = nullptr;
static int g_ldx = 0;
static std::vector<int32_t*> g_slotsets;
static Args g_rope;  // rope epilogue operands (positions, slots, tables, pools)
template <int CT, int WK, int DEPTH, int MODE, int XF = 0, int ORD = 0, int EPI = 0>
static float run(const Image& im, const std::vector<uint8_t*>& sets, const f16* dx, f16* dout, float* dslabs, int M, int S,
                 int iters, bool check, const std::vector<f16>& hx) {
    Args a;
    a.x = dx; a.ldx = g_ldx; a.xf = g_xf; a.out = dout; a.ldo = im.N; a.slabs = dslabs;
    a.M = M; a.K = im.K; a.N = im.N; a.NT = im.NT; a.KS = im.KS; a.G = im.G;
    a.offB = im.offB; a.S = S; a.steps = im.K / 64; a.trace = nullptr;
    a.positions = g_rope.positions; a.slots = g_rope.slots; a.cosb = g_rope.cosb; a.sinb = g_rope.sinb;
    a.kpool = g_rope.kpool; a.vpool = g_rope.vpool; a.rH = 32; a.rHkv = 32; a.rD = 128;
    {
        static unsigned* cnt = nullptr; static f16 *res = nullptr, *ho = nullptr, *hf = nullptr, *nw = nullptr; static float *sq = nullptr, *sqi = nullptr;
        if (!cnt) {
            CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMemset(cnt, 0, 4096 * 4));
            CK(hipMalloc(&res, 32 * 32768 * 2)); CK(hipMemset(res, 0, 32 * 32768 * 2));
            CK(hipMalloc(&ho, 32 * 32768 * 2)); CK(hipMalloc(&hf, 32 * 32768 * 2));
            CK(hipMalloc(&nw, 32768 * 2)); CK(hipMemset(nw, 0x3c, 32768 * 2));
            CK(hipMalloc(&sq, 32 * 1024 * 4)); CK(hipMalloc(&sqi, 32 * 128 * 4)); CK(hipMemset(sqi, 0x3c, 32 * 128 * 4));
        }
        static float* nsl = nullptr; static f16 *nr = nullptr, *nro = nullptr, *nx = nullptr; static unsigned* nf = nullptr;
        if (!nsl) {
            CK(hipMalloc(&nsl, (size_t)8 * 4 * 32 * 4096 * 4)); CK(hipMemset(nsl, 0, (size_t)8 * 4 * 32 * 4096 * 4));
            CK(hipMalloc(&nr, 32 * 4096 * 2)); CK(hipMemset(nr, 0x3c, 32 * 4096 * 2));
            CK(hipMalloc(&nro, 32 * 4096 * 2)); CK(hipMalloc(&nx, 32 * 4096 * 2)); CK(hipMemset(nx, 0, 32 * 4096 * 2));
            CK(hipMalloc(&nf, 256)); CK(hipMemset(nf, 0, 256));
        }
        a.nslabs = nsl; a.nres = nr; a.nres_out = nro; a.nxf = nx; a.nflag = nf; a.ntarget = 0;
        a.counters = cnt; a.resid = res; a.hout = ho; a.hfrag = hf; a.ssq = sq; a.normw = nw; a.ssq_in = sqi; a.ssq_n = 128; a.eps = 1e-5f;
    }
    int spg = im.gs / 64, sh = 0;
    while ((1 << sh) < spg) ++sh;
    a.spg_shift = sh;
    const int cgs = (im.NT + CT - 1) / CT;
    const size_t lds = (size_t)WK * CT * 4096;
    CK(hipFuncSetAttribute((const void*)wide_gemm<CT, WK, DEPTH, MODE, XF, ORD, false, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)wide_gemm<CT, WK, DEPTH, MODE, XF, ORD, true, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(cgs, S);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) {
            a.prep = sets[i % sets.size()];
            if (EPI >= 3 && getenv("WIDE_COLD")) a.slots = g_slotsets[i % g_slotsets.size()];
            if (EPI == 12 || EPI == 13) {
                static unsigned launches = 0;
                static const float* slab0 = a.nslabs;
                a.nslabs = slab0 + (size_t)(launches % 8) * 4 * 32 * 4096;   // slabs written "by the kernel before": not in this L2
                a.xf = a.nxf;
                if (EPI == 12) {
                    a.ntarget = 32u * (++launches);
                    hipLaunchKernelGGL((wide_gemm<CT, WK, DEPTH, MODE, XF, ORD, false, EPI>), dim3(cgs + 32, S), dim3(64 * WK), lds, 0, a);
                } else {
                    ++launches;
                    hipLaunchKernelGGL(norm_emul_kernel, dim3(32), dim3(512), 0, 0, a);
                    hipLaunchKernelGGL((wide_gemm<CT, WK, DEPTH, MODE, XF, ORD, false, 0>), grid, dim3(64 * WK), lds, 0, a);
                }
                continue;
            }
            hipLaunchKernelGGL((wide_gemm<CT, WK, DEPTH, MODE, XF, ORD, false, EPI>), grid, dim3(64 * WK), lds, 0, a);
        }
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 1000.f / iters);
    }
    if (getenv("WIDE_TRACE")) {
        const int nw = cgs * S * WK;
        long long* dtr; CK(hipMalloc(&dtr, (size_t)nw * 8 * 8)); CK(hipMemset(dtr, 0, (size_t)nw * 8 * 8));
        a.trace = dtr;
        for (int i = 0; i < 4; ++i) {   // the last of a few cold launches
            a.prep = sets[(i + 3) % sets.size()];
            hipLaunchKernelGGL((wide_gemm<CT, WK, DEPTH, MODE, XF, ORD, true, EPI>), grid, dim3(64 * WK), lds, 0, a);
        }
        CK(hipDeviceSynchronize());
        std::vector<long long> tr((size_t)nw * 8);
        CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
        long long rt0 = tr[7];
        for (int i = 0; i < nw; ++i) rt0 = std::min(rt0, tr[(size_t)i * 8 + 7]);
        {
            std::vector<long long> v(nw);
            for (int i = 0; i < nw; ++i) v[i] = (tr[(size_t)i * 8 + 7] - rt0) * 10;
            std::sort(v.begin(), v.end());
            printf("    wave entry after the first wave's entry (ns, s_memrealtime): median %lld  p90 %lld  max %lld\n", v[nw / 2], v[nw * 9 / 10], v[nw - 1]);
        }
        printf("    trace (s_memtime ticks from the wave's own entry; min / median / max over %d waves):\n", nw);
        const char* names[7] = {"entry", "prologue issued", "loop done", "first of last group", "last group done", "after barrier", "stored"};
        for (int k = 0; k < 7; ++k) {
            std::vector<long long> v(nw);
            for (int i = 0; i < nw; ++i) v[i] = tr[(size_t)i * 8 + k] - tr[(size_t)i * 8];
            std::sort(v.begin(), v.end());
            printf("      %-20s %7lld %7lld %7lld\n", names[k], v[0], v[nw / 2], v[nw - 1]);
        }
        a.trace = nullptr;
        CK(hipFree(dtr));
    }
    double maxerr = 0;
    if (check && MODE == 0 && EPI == 0) {
        a.prep = sets[0];
        CK(hipMemset(dout, 0, (size_t)32 * im.N * 2));
        hipLaunchKernelGGL((wide_gemm<CT, WK, DEPTH, MODE, XF, ORD, false, EPI>), grid, dim3(64 * WK), lds, 0, a);
        CK(hipDeviceSynchronize());
        std::vector<f16> ho((size_t)32 * im.N);
        std::vector<float> hs;
        if (S == 1) CK(hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost));
        else { hs.resize((size_t)S * 32 * im.NT * 32); CK(hipMemcpy(hs.data(), dslabs, hs.size() * 4, hipMemcpyDeviceToHost)); }
        for (int t = 0; t < 48; ++t) {
            const int m = rnd() % M, n = t < 4 ? (t & 1 ? im.N - 1 - (t >> 1) : (t >> 1)) : rnd() % im.N;
            double got;
            if (S == 1) got = (float)ho[(size_t)m * im.N + n];
            else { got = 0; for (int s2 = 0; s2 < S; ++s2) got += hs[((size_t)s2 * 32 + m) * (im.NT * 32) + n]; }
            const double want = ref_dot(im, hx, im.K, m, n);
            maxerr = std::max(maxerr, fabs(got - want) / (1.0 + fabs(want)));
        }
    }
    const int blocks = cgs * S;
    const double mb = (double)(im.offB * 1.0 * (im.K / 64) / im.KS + (double)im.NT * im.G * 128) / 1e6;
    printf("  CT %d WK %2d DEPTH %d S %2d mode %d xf %d ord %d epi %d ldx %d: blocks %4d  %6.2f us  %5.2f TB/s  relerr %.1e%s\n", CT, WK, DEPTH, S, MODE, XF, ORD, EPI, g_ldx, blocks, best,
           mb / best, maxerr, (check && MODE == 0 && maxerr > 3e-3) ? "  <-- WRONG" : "");
    return best;
}


// ---- round 5 runner: wide2_gemm, timed as back-to-back launches over rotating weight sets; WIDE_TRACE=1 adds the per-wave
// timeline, overall and by wave index (is the arrival skew systematic by wave age?)
template <int CT, int WK, int DW, int DX, int ORD = 1, int PRIO = 0, int PRE = 0, int KB = 0, int PB = 0, int EPI = 0, int AR = 0>
static float run2(const Image& im, const std::vector<uint8_t*>& sets, const f16* dx, f16* dout, float* dslabs, int M, int S,
                  int iters, const std::vector<f16>& hx) {
    Args a;
    memset(&a, 0, sizeof(a));
    a.x = dx; a.ldx = im.K; a.xf = g_xf; a.out = dout; a.ldo = im.N; a.slabs = dslabs;
    a.M = M; a.K = im.K; a.N = im.N; a.NT = im.NT; a.KS = im.KS; a.G = im.G;
    a.offB = im.offB; a.S = S; a.steps = im.K / 64; a.trace = nullptr;
    a.lay = getenv("WIDE_LAY") ? atoi(getenv("WIDE_LAY")) : 0;
    int spg = im.gs / 64, sh = 0;
    while ((1 << sh) < spg) ++sh;
    a.spg_shift = sh;
    const int cgs = (im.NT + CT - 1) / CT;
    const size_t lds = (size_t)WK * CT * 4096 + (EPI == 22 ? 32 * CT * 16 * 2 : 0);
    auto kern = wide2_gemm<CT, WK, DW, DX, ORD, PRIO, PRE, false, KB, PB, EPI, AR>;
    auto kern_tr = wide2_gemm<CT, WK, DW, DX, ORD, PRIO, PRE, true, KB, PB, EPI, AR>;
    if (AR && ((im.K / 64 / S) % (2 * WK) != 0 || im.gs != 128 || KB)) { printf("  (lean arithmetic needs whole groups per wave: skipped)\n"); return 0.f; }
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)kern_tr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 64 * WK, lds));
    dim3 grid(cgs, S);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) {
            a.prep = sets[i % sets.size()];
            hipLaunchKernelGGL(kern, grid, dim3(64 * WK), lds, 0, a);
        }
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 1000.f / iters);
    }
    // check
    double maxerr = 0;
    if (EPI == 0) {
        a.prep = sets[0];
        CK(hipMemset(dout, 0, (size_t)32 * im.N * 2));
        hipLaunchKernelGGL(kern, grid, dim3(64 * WK), lds, 0, a);
        CK(hipDeviceSynchronize());
        std::vector<f16> ho((size_t)32 * im.N);
        std::vector<float> hs;
        if (S == 1) CK(hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost));
        else { hs.resize((size_t)S * 32 * im.NT * 32); CK(hipMemcpy(hs.data(), dslabs, hs.size() * 4, hipMemcpyDeviceToHost)); }
        for (int t = 0; t < 32; ++t) {
            const int m = rnd() % M, n = t < 4 ? (t & 1 ? im.N - 1 - (t >> 1) : (t >> 1)) : rnd() % im.N;
            double got;
            if (S == 1) got = (float)ho[(size_t)m * im.N + n];
            else { got = 0; for (int s2 = 0; s2 < S; ++s2) got += hs[((size_t)s2 * 32 + m) * (im.NT * 32) + n]; }
            const double want = ref_dot(im, hx, im.K, m, n);
            maxerr = std::max(maxerr, fabs(got - want) / (1.0 + fabs(want)));
        }
    }
    const int blocks = cgs * S;
    const double mb = (double)(im.offB * 1.0 * (im.K / 64) / im.KS + (double)im.NT * im.G * 128) / 1e6;
    if (AR) printf("  [lean]");
    printf("  CT %d WK %2d DW %d DX %d S %2d ord %d prio %d pre %d kb %d pb %d epi %2d: blocks %4d (%d/CU fit)  %6.2f us  %5.2f TB/s  relerr %.1e%s\n", CT, WK, DW, DX, S, ORD,
           PRIO, PRE, KB, PB, EPI, blocks, occ, best, mb / best, maxerr, maxerr > 3e-3 ? "  <-- WRONG" : "");
    if (getenv("WIDE_TRACE")) {
        const int nw = cgs * S * WK;
        long long* dtr; CK(hipMalloc(&dtr, (size_t)nw * 8 * 8)); CK(hipMemset(dtr, 0, (size_t)nw * 8 * 8));
        a.trace = dtr;
        for (int i = 0; i < 4; ++i) {
            a.prep = sets[(i + 3) % sets.size()];
            hipLaunchKernelGGL(kern_tr, grid, dim3(64 * WK), lds, 0, a);
        }
        CK(hipDeviceSynchronize());
        std::vector<long long> tr((size_t)nw * 8);
        CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
        long long rt0 = tr[7];
        for (int i = 0; i < nw; ++i) rt0 = std::min(rt0, tr[(size_t)i * 8 + 7]);
        {
            std::vector<long long> v(nw);
            for (int i = 0; i < nw; ++i) v[i] = (tr[(size_t)i * 8 + 7] - rt0) * 10;
            std::sort(v.begin(), v.end());
            printf("    wave entry after the first wave's entry (ns): median %lld  p90 %lld  max %lld\n", v[nw / 2], v[nw * 9 / 10], v[nw - 1]);
        }
        const char* names[7] = {"entry", "prologue issued", "loop done", "first of last group", "last group done", "after barrier", "stored"};
        printf("    ticks from the wave's own entry, min / median / max over %d waves; then medians by wave index 0..%d\n", nw, WK - 1);
        for (int k = 1; k < 7; ++k) {
            std::vector<long long> v(nw);
            for (int i = 0; i < nw; ++i) v[i] = tr[(size_t)i * 8 + k] - tr[(size_t)i * 8];
            std::sort(v.begin(), v.end());
            printf("      %-20s %6lld %6lld %6lld |", names[k], v[0], v[nw / 2], v[nw - 1]);
            for (int w = 0; w < WK; ++w) {
                std::vector<long long> u;
                for (int i = w; i < nw; i += WK) u.push_back(tr[(size_t)i * 8 + k] - tr[(size_t)i * 8]);
                std::sort(u.begin(), u.end());
                printf(" %6lld", u[u.size() / 2]);
            }
            printf("\n");
        }
        CK(hipFree(dtr));
    }
    return best;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32;
    {   // rope operands at the cfg3 shape: 32 rows, 32 + 32 + 32 heads of 128, one page of 32 tokens per (page, head)
        std::vector<int32_t> pos(32), slots(32);
        const int pages = 64 * 32;
        for (int i = 0; i < 32; ++i) { pos[i] = 1000 + (rnd() % 48); slots[i] = (int)((rnd() % pages) * 32 + rnd() % 32); }
        int32_t *dp, *ds; CK(hipMalloc(&dp, 128)); CK(hipMalloc(&ds, 128));
        CK(hipMemcpy(dp, pos.data(), 128, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, slots.data(), 128, hipMemcpyHostToDevice));
        f16 *dc, *dsn, *kp, *vp;
        CK(hipMalloc(&dc, 2048 * 64 * 2)); CK(hipMalloc(&dsn, 2048 * 64 * 2)); CK(hipMemset(dc, 0, 2048 * 64 * 2)); CK(hipMemset(dsn, 0, 2048 * 64 * 2));
        CK(hipMalloc(&kp, (size_t)pages * 32 * 32 * 128 * 2)); CK(hipMalloc(&vp, (size_t)pages * 32 * 32 * 128 * 2));
        for (int k = 0; k < 64; ++k) {
            std::vector<int32_t> sl(32);
            for (int i = 0; i < 32; ++i) sl[i] = (int)((rnd() % pages) * 32 + rnd() % 32);
            int32_t* d; CK(hipMalloc(&d, 128)); CK(hipMemcpy(d, sl.data(), 128, hipMemcpyHostToDevice));
            g_slotsets.push_back(d);
        }
        g_rope.positions = dp; g_rope.slots = ds; g_rope.cosb = dc; g_rope.sinb = dsn; g_rope.kpool = kp; g_rope.vpool = vp;
    }
    struct Shape { const char* name; int K, N; } shapes[] = {
        {"qkv 4096x12288", 4096, 12288}, {"o 4096x4096", 4096, 4096}, {"gate_up 4096x22016", 4096, 22016}, {"down 11008x4096", 11008, 4096}};
    for (auto& sh : shapes) {
        Image im = make_image(sh.K, sh.N, 128);
        const int nsets = getenv("WIDE_ONESET") ? 1 : (int)std::max<int64_t>(2, (700ll << 20) / im.total);
        std::vector<uint8_t*> sets(nsets);
        for (int i = 0; i < nsets; ++i) {
            CK(hipMalloc(&sets[i], im.total));
            CK(hipMemcpy(sets[i], im.host.data(), im.total, hipMemcpyHostToDevice));
        }
        std::vector<f16> hx((size_t)32 * sh.K);
        for (auto& v : hx) v = (f16)(((int)(rnd() % 2001) - 1000) / 1000.f);
        f16* dx; CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
        f16* dxp; CK(hipMalloc(&dxp, (size_t)32 * (sh.K + 64) * 2));
        CK(hipMemcpy2D(dxp, (size_t)(sh.K + 64) * 2, hx.data(), (size_t)sh.K * 2, (size_t)sh.K * 2, 32, hipMemcpyHostToDevice));
        std::vector<f16> hxf(hx.size());
        for (int st = 0; st < sh.K / 64; ++st)
            for (int i = 0; i < 4; ++i)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e)
                        hxf[(((size_t)st * 4 + i) * 64 + l) * 8 + e] = hx[(size_t)(l & 31) * sh.K + st * 64 + (l >> 5) * 32 + i * 8 + e];
        f16* dxf; CK(hipMalloc(&dxf, hxf.size() * 2)); CK(hipMemcpy(dxf, hxf.data(), hxf.size() * 2, hipMemcpyHostToDevice));
        g_xf = dxf;
        f16* dout; CK(hipMalloc(&dout, (size_t)32 * sh.N * 2));
        float* dslabs; CK(hipMalloc(&dslabs, (size_t)16 * 32 * im.NT * 32 * 4));
        printf("%s  (%.1f MB image, %d rotating sets, M = %d)\n", sh.name, im.total / 1e6, nsets, M);
        const int it = getenv("WIDE_ONESET") ? 40 : 2 * nsets;
#define R(CT, WK, D, S) (g_ldx = sh.K, run<CT, WK, D, 0, 0, 1>(im, sets, dx, dout, dslabs, M, S, it, true, hx))
#define RP(CT, WK, D, S) (g_ldx = sh.K + 64, run<CT, WK, D, 0, 0, 1>(im, sets, dxp, dout, dslabs, M, S, it, true, hx))
#define RX(CT, WK, D, S) (g_ldx = sh.K, run<CT, WK, D, 0, 1, 1>(im, sets, dx, dout, dslabs, M, S, it, true, hx))
#define RX0(CT, WK, D, S) (g_ldx = sh.K, run<CT, WK, D, 0, 1, 0>(im, sets, dx, dout, dslabs, M, S, it, true, hx))
#define RE(CT, WK, D, S, EPI) (g_ldx = sh.K, run<CT, WK, D, 0, 1, 1, EPI>(im, sets, dx, dout, dslabs, M, S, it, false, hx))
#define RM(CT, WK, D, S, MODE) (g_ldx = sh.K, run<CT, WK, D, MODE, 1, 1>(im, sets, dx, dout, dslabs, M, S, it, false, hx))
#define R2(CT, WK, DW, DX, S, ...) run2<CT, WK, DW, DX, ##__VA_ARGS__>(im, sets, dx, dout, dslabs, M, S, it, hx)
        const char* suite = getenv("WIDE_SUITE") ? getenv("WIDE_SUITE") : "r04";
        if (!strcmp(suite, "r05a")) {
            // the shipped plan of each shape first (DW = DX = 2, ord 1), then one change at a time
            if (sh.N == 12288) {
                R2(2, 8, 2, 2, 1); R2(2, 8, 2, 2, 1, 1, 0, 1); R2(2, 8, 2, 2, 1, 2); R2(2, 8, 2, 2, 1, 1, 1); R2(2, 8, 2, 2, 1, 1, 2);
                R2(2, 8, 4, 2, 1); R2(2, 8, 4, 2, 1, 2); R2(2, 8, 3, 1, 1); R2(2, 8, 4, 1, 1); R2(2, 8, 2, 1, 1);
                R2(2, 4, 2, 2, 1); R2(2, 4, 4, 2, 1); R2(1, 4, 4, 2, 1); R2(1, 4, 2, 2, 1); R2(1, 8, 4, 2, 1); R2(3, 8, 4, 2, 1); R2(3, 4, 4, 2, 1);
            } else if (sh.N == 22016) {
                R2(3, 8, 2, 2, 1); R2(3, 8, 2, 2, 1, 1, 0, 1); R2(3, 8, 2, 2, 1, 2); R2(3, 8, 2, 2, 1, 1, 1); R2(3, 8, 2, 2, 1, 1, 2);
                R2(3, 8, 4, 2, 1); R2(3, 8, 4, 2, 1, 2); R2(3, 8, 3, 1, 1); R2(3, 8, 4, 1, 1);
                R2(2, 4, 2, 2, 1); R2(2, 4, 4, 2, 1); R2(3, 4, 4, 2, 1); R2(2, 8, 4, 2, 1); R2(1, 4, 4, 2, 1); R2(4, 8, 4, 2, 1);
            } else if (sh.K == 4096) {   // o
                R2(2, 8, 2, 2, 4); R2(2, 8, 2, 2, 4, 1, 0, 1); R2(2, 8, 2, 2, 4, 2); R2(2, 8, 2, 2, 4, 1, 1);
                R2(2, 8, 4, 2, 4); R2(2, 8, 2, 1, 4); R2(2, 8, 2, 2, 2); R2(2, 8, 4, 2, 2);
                R2(2, 4, 2, 2, 4); R2(2, 4, 4, 2, 4); R2(2, 4, 2, 2, 8); R2(2, 4, 4, 2, 8); R2(1, 4, 2, 2, 4); R2(1, 4, 4, 2, 4); R2(1, 4, 4, 2, 2); R2(1, 8, 4, 2, 2);
            } else {   // down
                R2(2, 8, 2, 2, 4); R2(2, 8, 2, 2, 4, 1, 0, 1); R2(2, 8, 2, 2, 4, 2); R2(2, 8, 2, 2, 4, 1, 1);
                R2(2, 8, 4, 2, 4); R2(2, 8, 4, 2, 4, 2); R2(2, 8, 2, 1, 4); R2(2, 8, 4, 2, 2); R2(2, 8, 4, 2, 3);
                R2(2, 4, 2, 2, 4); R2(2, 4, 4, 2, 4); R2(2, 4, 4, 2, 8); R2(2, 4, 4, 2, 6); R2(1, 4, 4, 2, 4); R2(1, 4, 4, 2, 2); R2(1, 8, 4, 2, 2);
            }
        } else if (!strcmp(suite, "r05b")) {
            // the weighted k split (kb), with / without the argument prefetch, the priority of the younger half, the prologue barrier
            if (sh.N == 12288) {
                R2(2, 8, 2, 2, 1); R2(2, 8, 2, 2, 1, 1, 0, 0, 1); R2(2, 8, 2, 2, 1, 1, 0, 0, 2); R2(2, 8, 2, 2, 1, 1, 0, 0, 3);
                R2(2, 8, 2, 2, 1, 1, 0, 1, 1); R2(2, 8, 2, 2, 1, 1, 0, 1, 2); R2(2, 8, 2, 2, 1, 1, 1, 1, 1); R2(2, 8, 2, 2, 1, 1, 1, 1, 2);
                R2(2, 8, 2, 2, 1, 1, 0, 1, 0, 1); R2(2, 8, 2, 2, 1, 1, 0, 1, 1, 1); R2(2, 8, 2, 2, 1, 1, 0, 1, 2, 1);
            } else if (sh.N == 22016) {
                R2(3, 8, 2, 2, 1); R2(3, 8, 2, 2, 1, 1, 0, 0, 1); R2(3, 8, 2, 2, 1, 1, 0, 0, 2); R2(3, 8, 2, 2, 1, 1, 0, 0, 3);
                R2(3, 8, 2, 2, 1, 1, 0, 1, 1); R2(3, 8, 2, 2, 1, 1, 0, 1, 2); R2(3, 8, 2, 2, 1, 1, 1, 1, 1); R2(3, 8, 2, 2, 1, 1, 1, 1, 2);
                R2(3, 8, 2, 2, 1, 1, 0, 1, 0, 1); R2(3, 8, 2, 2, 1, 1, 0, 1, 1, 1); R2(3, 8, 2, 2, 1, 1, 0, 1, 2, 1);
                // SiLU * up epilogues on the best split so far and on the plain one
                R2(3, 8, 2, 2, 1, 1, 0, 1, 0, 0, 20); R2(3, 8, 2, 2, 1, 1, 0, 1, 0, 0, 21); R2(3, 8, 2, 2, 1, 1, 0, 1, 0, 0, 22);
                R2(3, 8, 2, 2, 1, 1, 0, 1, 1, 0, 21); R2(3, 8, 2, 2, 1, 1, 0, 1, 1, 0, 22);
            } else if (sh.K == 4096) {   // o
                R2(2, 8, 2, 2, 4); R2(2, 8, 2, 2, 4, 1, 0, 0, 1); R2(2, 8, 2, 2, 4, 1, 0, 1, 1); R2(2, 8, 2, 2, 4, 1, 1, 1, 1);
                R2(2, 8, 2, 2, 4, 1, 0, 1, 0, 1); R2(2, 8, 2, 2, 4, 1, 0, 1, 1, 1); R2(2, 8, 2, 2, 2, 1, 0, 1, 1); R2(2, 8, 2, 2, 2, 1, 0, 1, 2);
                R2(2, 8, 2, 2, 3, 1, 0, 1, 1);
            } else {   // down
                R2(2, 8, 2, 2, 4); R2(2, 8, 2, 2, 4, 1, 0, 0, 1); R2(2, 8, 2, 2, 4, 1, 0, 0, 2); R2(2, 8, 2, 2, 4, 1, 0, 1, 1); R2(2, 8, 2, 2, 4, 1, 0, 1, 2);
                R2(2, 8, 2, 2, 4, 1, 1, 1, 1); R2(2, 8, 2, 2, 4, 1, 0, 1, 0, 1); R2(2, 8, 2, 2, 4, 1, 0, 1, 1, 1); R2(2, 8, 2, 2, 3, 1, 0, 1, 1); R2(2, 8, 2, 2, 3, 1, 0, 1, 2);
                R2(2, 8, 2, 2, 2, 1, 0, 1, 2); R2(2, 8, 2, 2, 2, 1, 0, 1, 3);
            }
        } else if (!strcmp(suite, "r05l")) {   // the shipped plans; run with WIDE_LAY=0 / 1 / 2 (weight address order, timing only)
            if (sh.N == 12288) { R2(2, 8, 2, 2, 1, 1, 0, 1); R2(2, 8, 4, 2, 1, 1, 0, 1); }
            else if (sh.N == 22016) { R2(3, 8, 2, 2, 1, 1, 0, 1); R2(3, 8, 4, 2, 1, 1, 0, 1); }
            else { R2(2, 8, 2, 2, 4, 1, 0, 1); R2(2, 8, 4, 2, 4, 1, 0, 1); R2(2, 8, 2, 2, 2, 1, 0, 1); }
        } else if (!strcmp(suite, "r05d")) {   // what bounds the loop?  ablations of the round-4 kernel (wrong results by design)
            // 0 full; 1 no arithmetic; 2 no x loads; 6 no dequantisation; 5 weights + scales only; 4 weights only;
            // 8 x always from the same 4 KiB (L1 hits); 9 one x load per step instead of four
            if (sh.N == 12288) {
                RM(2, 8, 2, 1, 0); RM(2, 8, 2, 1, 1); RM(2, 8, 2, 1, 2); RM(2, 8, 2, 1, 6); RM(2, 8, 2, 1, 5); RM(2, 8, 2, 1, 4); RM(2, 8, 2, 1, 8); RM(2, 8, 2, 1, 9);
            } else if (sh.N == 22016) {
                RM(3, 8, 2, 1, 0); RM(3, 8, 2, 1, 1); RM(3, 8, 2, 1, 2); RM(3, 8, 2, 1, 6); RM(3, 8, 2, 1, 5); RM(3, 8, 2, 1, 4); RM(3, 8, 2, 1, 8); RM(3, 8, 2, 1, 9);
            } else if (sh.K == 4096) {
                RM(2, 8, 2, 4, 0); RM(2, 8, 2, 4, 1); RM(2, 8, 2, 4, 2); RM(2, 8, 2, 4, 5); RM(2, 8, 2, 4, 4); RM(2, 8, 2, 4, 8); RM(2, 8, 2, 4, 9);
            } else {
                RM(2, 8, 2, 4, 0); RM(2, 8, 2, 4, 1); RM(2, 8, 2, 4, 2); RM(2, 8, 2, 4, 5); RM(2, 8, 2, 4, 4); RM(2, 8, 2, 4, 8); RM(2, 8, 2, 4, 9);
            }
        } else if (!strcmp(suite, "r05c")) {   // lean arithmetic against the exact dequantisation, same skeleton
            if (sh.N == 12288) {
                R2(2, 8, 2, 2, 1, 1, 0, 1); R2(2, 8, 2, 2, 1, 1, 0, 1, 0, 0, 0, 1); R2(3, 8, 2, 2, 1, 1, 0, 1, 0, 0, 0, 1); R2(2, 4, 2, 2, 1, 1, 0, 1, 0, 0, 0, 1);
            } else if (sh.N == 22016) {
                R2(3, 8, 2, 2, 1, 1, 0, 1); R2(3, 8, 2, 2, 1, 1, 0, 1, 0, 0, 0, 1); R2(4, 8, 2, 2, 1, 1, 0, 1, 0, 0, 0, 1); R2(2, 8, 2, 2, 1, 1, 0, 1, 0, 0, 0, 1);
                R2(3, 8, 2, 2, 1, 1, 0, 1, 0, 0, 21); R2(3, 8, 2, 2, 1, 1, 0, 1, 0, 0, 21, 1);
            } else if (sh.K == 4096) {
                R2(2, 8, 2, 2, 4, 1, 0, 1); R2(2, 8, 2, 2, 4, 1, 0, 1, 0, 0, 0, 1); R2(2, 8, 2, 2, 2, 1, 0, 1, 0, 0, 0, 1);
            } else {
                R2(2, 8, 2, 2, 4, 1, 0, 1);
            }
        } else if (!strcmp(suite, "r05t")) {   // run with WIDE_TRACE=1: timelines of the shipped plans and of the deeper weight ring
            if (getenv("WIDE_T3")) {   // third trace set: lean arithmetic
                if (sh.N == 12288) { R2(2, 8, 2, 2, 1, 1, 0, 1, 0, 0, 0, 1); }
                else if (sh.N == 22016) { R2(3, 8, 2, 2, 1, 1, 0, 1, 0, 0, 0, 1); }
                else if (sh.K == 4096) { R2(2, 8, 2, 2, 4, 1, 0, 1, 0, 0, 0, 1); }
            }
            else if (getenv("WIDE_T2")) {   // second trace set: the weighted split
                if (sh.N == 12288) { R2(2, 8, 2, 2, 1, 1, 0, 1, 1); R2(2, 8, 2, 2, 1, 1, 0, 1, 2); }
                else if (sh.N == 22016) { R2(3, 8, 2, 2, 1, 1, 0, 1, 1); R2(3, 8, 2, 2, 1, 1, 0, 1, 2); R2(3, 8, 2, 2, 1, 1, 0, 1, 1, 0, 22); }
                else { R2(2, 8, 2, 2, 4, 1, 0, 1, 1); }
            }
            else if (sh.N == 12288) { R2(2, 8, 2, 2, 1); R2(2, 8, 4, 2, 1); R2(2, 4, 4, 2, 1); }
            else if (sh.N == 22016) { R2(3, 8, 2, 2, 1); R2(3, 8, 4, 2, 1); R2(2, 4, 4, 2, 1); }
            else { R2(2, 8, 2, 2, 4); R2(2, 8, 4, 2, 4); R2(2, 4, 4, 2, 4); }
        } else if (sh.N == 12288) {
            RX(2, 8, 2, 1); RE(2, 8, 2, 1, 13); RE(2, 8, 2, 1, 12);
        } else if (sh.N == 22016) {
            RX(3, 8, 2, 1); RE(3, 8, 2, 1, 13); RE(3, 8, 2, 1, 12);
        }
        for (auto p : sets) CK(hipFree(p));
        CK(hipFree(dx)); CK(hipFree(dout)); CK(hipFree(dslabs));
    }
    return 0;
}
