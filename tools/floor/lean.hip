// Calibration of the "lean" int4 step (round 3): the nibbles go to the MFMA as f16 (1024 + q) / (64 + q) — one
// v_and_or per pair, no subtraction, no scale — a per-group accumulator takes the 8 MFMAs of a 128-row group plus ONE
// correction MFMA whose A operand holds the split row sums of x (so the offsets and the zero point cancel exactly), and
// the scale is applied when the group is folded into the output accumulator (16 FMAs per group).
// Variants: 0 = old step (dequantise fully, A from LDS: compute.hip mode 3), 1 = lean unpack + MFMA only (no fold),
// 2 = lean with correction MFMA + fold, A and A' from LDS, 3 = variant 2 with the weights read from an LDS ring too.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned and_or(unsigned q, unsigned mask, unsigned ex) {
    unsigned r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(q), "s"(mask), "v"(ex));
    return r;
}
__device__ __forceinline__ f16x8 dequant8(unsigned q, f16x2 zc, f16x2 zd, f16x2 sc, unsigned EX, unsigned M0, unsigned M1) {
    const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
    unsigned q2 = q >> 8;
    unsigned a0 = and_or(q, M0, EX), a1 = and_or(q, M1, EX), a2 = and_or(q2, M0, EX), a3 = and_or(q2, M1, EX);
    f16x2 h0 = (__builtin_bit_cast(f16x2, a0) - zc) * sc;
    f16x2 h1 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a1), r16, zd) * sc;
    f16x2 h2 = (__builtin_bit_cast(f16x2, a2) - zc) * sc;
    f16x2 h3 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a3), r16, zd) * sc;
    u32x4 p = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, h2),
               __builtin_bit_cast(unsigned, h3)};
    return __builtin_bit_cast(f16x8, p);
}
// 8 nibbles -> (1024+n0, 1024+n4 | 64+n1, 64+n5 | 1024+n2, 1024+n6 | 64+n3, 64+n7): 5 VALU
__device__ __forceinline__ f16x8 unpack8(unsigned q, unsigned EXA, unsigned EXB, unsigned M0, unsigned M1) {
    unsigned q2 = q >> 8;
    u32x4 p = {and_or(q, M0, EXA), and_or(q, M1, EXB), and_or(q2, M0, EXA), and_or(q2, M1, EXB)};
    return __builtin_bit_cast(f16x8, p);
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

template <int MODE>
__global__ void step_kernel(const unsigned* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    u32x4 q = *reinterpret_cast<const u32x4*>(in + (threadIdx.x & 255) * 4);
    unsigned EX = 0x64006400u, EXB = 0x54005400u, M0 = 0x000F000Fu, M1 = 0x00F000F0u;
    asm volatile("" : "+v"(EX), "+v"(EXB));
    asm volatile("" : "+s"(M0), "+s"(M1));
    f16x2 zc = {(f16)1032.f, (f16)1032.f}, zd = {(f16)-72.f, (f16)-72.f}, sc = {(f16)0.01f, (f16)0.01f};
    f16x8 av = __builtin_bit_cast(f16x8, q);
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc0 = zero, acc1 = zero;
    __shared__ __attribute__((aligned(16))) f16 xs[32 * 264];
    __shared__ __attribute__((aligned(16))) f16 xsum[2 * 64 * 8];      // A' fragments of two groups
    __shared__ __attribute__((aligned(16))) unsigned wring[16 * 4 * 256];  // 4 one-KiB slots per wave
    for (int i = threadIdx.x; i < 32 * 264 / 8; i += blockDim.x) reinterpret_cast<f16x8*>(xs)[i] = av;
    for (int i = threadIdx.x; i < 2 * 64; i += blockDim.x) reinterpret_cast<f16x8*>(xsum)[i] = av;
    for (int i = threadIdx.x; i < 16 * 4 * 64; i += blockDim.x) reinterpret_cast<u32x4*>(wring)[i] = q;
    __syncthreads();
    const f16* xk = xs + (lane & 31) * 264 + (lane >> 5) * 32;
    const u32x4* wr = reinterpret_cast<const u32x4*>(wring) + w * 256 + lane;
    unsigned szw = in[lane & 31];  // {scale f16, zp1 f16} of this lane's column
    long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        f16x8 b[4];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = dequant8(q[i], zc, zd, sc, EX, M0, M1);
            q[0] += 0x11111111u * it;
            q[1] ^= q[0]; q[2] += q[1]; q[3] ^= q[2];
            __builtin_amdgcn_sched_barrier(0);
            const f16* xc = xk + (it & 3) * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f16x8 a2 = *reinterpret_cast<const f16x8*>(xc + i * 8);
                acc0 = MFMA(a2, b[i], acc0);
            }
        }
    } else if (MODE == 1) {
        f16x8 b[4];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = unpack8(q[i], EX, EXB, M0, M1);
            q[0] += 0x11111111u * it;
            q[1] ^= q[0]; q[2] += q[1]; q[3] ^= q[2];
            const f16* xc = xk + (it & 3) * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f16x8 a2 = *reinterpret_cast<const f16x8*>(xc + i * 8);
                acc0 = MFMA(a2, b[i], acc0);
            }
        }
    } else {
        // one iteration = two groups of 128 rows = 4 steps; group accumulators g0 / g1 alternate
        for (int it = 0; it < iters; it += 4) {
            f32x16 g[2];
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                // correction MFMA first (C = 0 starts the group): A' = split row sums, B' = (-1,-1,-1,-zp1,-zp1,-zp1,0,0)
                const f16x2 szh = __builtin_bit_cast(f16x2, szw);
                const f16 nz = -szh[1];
                const f16x8 bp = {(f16)-1.f, (f16)-1.f, (f16)-1.f, nz, nz, nz, (f16)0.f, (f16)0.f};
                const f16x8 ap = *reinterpret_cast<const f16x8*>(xsum + (grp * 64 + lane) * 8);
                g[grp] = MFMA(ap, bp, zero);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 cur = q;
                    if (MODE == 3) cur = wr[((grp * 2 + s) & 3) * 64];
                    f16x8 b[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) b[i] = unpack8(cur[i], EX, EXB, M0, M1);
                    if (MODE != 3) {
                        q[0] += 0x11111111u * it;
                        q[1] ^= q[0]; q[2] += q[1]; q[3] ^= q[2];
                    }
                    const f16* xc = xk + ((grp * 2 + s) & 3) * 64;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f16x8 a2 = *reinterpret_cast<const f16x8*>(xc + i * 8);
                        g[grp] = MFMA(a2, b[i], g[grp]);
                    }
                }
                szw += 0x00010001u;
            }
            // fold: out += s * g  (s per lane/column)
            const f16x2 szh = __builtin_bit_cast(f16x2, szw);
            const float s0 = (float)szh[0], s1 = (float)szh[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = __builtin_fmaf(s0, g[0][r], acc0[r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = __builtin_fmaf(s1, g[1][r], acc0[r]);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    f32x16 acc = acc0 + acc1;
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[(long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> void run(const char* name, int wps, unsigned* in, float* out, long long* cyc) {
    const int iters = 2000, threads = 64 * 4 * wps, blocks = 256;
    hipLaunchKernelGGL(step_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(step_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    long long h[16]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-22s waves/SIMD=%d: %8.1f cycles per wave-step (wave 0), %8.1f ns per SIMD-step-round (%.1f ns per wave-step per SIMD)\n",
           name, wps, (double)h[0] / iters, ms * 1e6 / iters, ms * 1e6 / iters / wps);
}

int main() {
    unsigned* in; CK(hipMalloc(&in, 4096)); CK(hipMemset(in, 0x5a, 4096));
    float* out; CK(hipMalloc(&out, 256 * 1024 * 4));
    long long* cyc; CK(hipMalloc(&cyc, 256 * 16 * 8));
    for (int wps = 1; wps <= 4; ++wps) {
        run<0>("old: dequant+lds+mfma", wps, in, out, cyc);
        run<1>("lean unpack+lds+mfma", wps, in, out, cyc);
        run<2>("lean +corr+fold", wps, in, out, cyc);
        run<3>("lean +corr+fold+wlds", wps, in, out, cyc);
    }
    return 0;
}
