// Calibration: issue-bound cost of one GPTQ k64-step (4 x dequant8 + 4 x MFMA 32x32x16) per wave, at 1..4 waves/SIMD,
// with operands in registers (no memory, no LDS).  Variants: dequant only, MFMA only, both.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// MODE 5: the nibbles go to the MFMA as f16 subnormals (q * 2^-24: one shift + one AND per pair, no zero point, no
// scale), the step accumulates into a per-group accumulator, and every second step (group of 128 rows) the group is
// folded into the output accumulator: acc += (s * 2^24) * tacc - (s * (z + 1)) * rowsum_x (rowsums from LDS).
template <int MODE>  // 0 both, 1 dequant only, 2 mfma only, 3 both + A fragments from LDS, 4 = 3 with A read one step ahead
__global__ void step_kernel(const unsigned* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 q = *reinterpret_cast<const u32x4*>(in + (threadIdx.x & 255) * 4);
    unsigned EX = 0x64006400u, M0 = 0x000F000Fu, M1 = 0x00F000F0u;
    asm volatile("" : "+v"(EX));
    asm volatile("" : "+s"(M0), "+s"(M1));
    f16x2 zc = {(f16)1032.f, (f16)1032.f}, zd = {(f16)-72.f, (f16)-72.f}, sc = {(f16)0.01f, (f16)0.01f};
    f16x8 av = __builtin_bit_cast(f16x8, q);
    f32x16 acc0 = {0}, acc1 = {0}, g0 = {0}, g1 = {0};
    f16x8 b[4];
    for (int i = 0; i < 4; ++i) b[i] = av;
    __shared__ __attribute__((aligned(16))) f16 xs[32 * 264];
    for (int i = threadIdx.x; i < 32 * 264 / 8; i += blockDim.x) reinterpret_cast<f16x8*>(xs)[i] = av;
    __syncthreads();
    const f16* xk = xs + (lane & 31) * 264 + (lane >> 5) * 32;
    f16x8 an[4];
    if (MODE == 4)
        for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const f16x8*>(xk + i * 8);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4) {
            f16x8 ac[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ac[i] = an[i];
            const f16* xn = xk + ((it + 1) & 3) * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const f16x8*>(xn + i * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = dequant8(q[i], zc, zd, sc, EX, M0, M1);
            q[0] += 0x11111111u * it;
            q[1] ^= q[0]; q[2] += q[1]; q[3] ^= q[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac[i], b[i], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac[i], b[i], acc0, 0, 0, 0);
            }
            continue;
        }
        if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned w = q[i];
                u32x4 p = {w & M0, (w >> 4) & M0, (w >> 8) & M0, (w >> 12) & M0};
                b[i] = __builtin_bit_cast(f16x8, p);
            }
            q[0] += 0x11111111u * it;
            q[1] ^= q[0]; q[2] += q[1]; q[3] ^= q[2];
            __builtin_amdgcn_sched_barrier(0);
            const f16* xc = xk + (it & 3) * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f16x8 a2 = *reinterpret_cast<const f16x8*>(xc + i * 8);
                if (i & 1) g1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b[i], g1, 0, 0, 0);
                else g0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b[i], g0, 0, 0, 0);
            }
            if (it & 1) {  // group boundary: fold and clear
                const float s24 = (float)sc[0] * 16777216.f + (float)(it & 2), sz = (float)zc[0] * (float)sc[0];
                const float* rs = reinterpret_cast<const float*>(xs) + (lane >> 5) * 4 + (it & 4);
                f32x16 t = g0 + g1;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 rr = *reinterpret_cast<const f32x4*>(rs + r4 * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc0[r4 * 4 + e] += s24 * t[r4 * 4 + e] - sz * rr[e];
                }
                g0 = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                g1 = g0;
            }
            continue;
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = dequant8(q[i], zc, zd, sc, EX, M0, M1);
            q[0] += 0x11111111u * it;
            q[1] ^= q[0]; q[2] += q[1]; q[3] ^= q[2];
            __builtin_amdgcn_sched_barrier(0);
            const f16* xc = xk + (it & 3) * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f16x8 a2 = *reinterpret_cast<const f16x8*>(xc + i * 8);
                if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b[i], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b[i], acc0, 0, 0, 0);
            }
            continue;
        }
        if (MODE != 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = dequant8(q[i], zc, zd, sc, EX, M0, M1);
            q[0] += 0x11111111u * it;  // keep the dequant loop-variant
            q[1] ^= q[0]; q[2] += q[1]; q[3] ^= q[2];
        }
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b[i], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b[i], acc0, 0, 0, 0);
            }
        } else {
            asm volatile("" ::"v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    f32x16 acc = acc0 + acc1;
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (MODE == 1) s += (float)b[0][0] + (float)b[3][7];
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
    printf("%-14s waves/SIMD=%d: %8.1f cycles per wave-step (wave 0), %8.1f ns per SIMD-step (all waves: %.1f ns per wave-step-slot)\n",
           name, wps, (double)h[0] / iters, ms * 1e6 / iters, ms * 1e6 / iters / wps);
}

__global__ void denorm_check(float* out) {
    // A = 1.0 everywhere, B = f16 subnormal with bits 0x0003 (3 * 2^-24): D[m][n] must be 16 * 3 * 2^-24
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (f16)1.0f;
    u32x4 bits = {0x00030003u, 0x00030003u, 0x00030003u, 0x00030003u};
    b = __builtin_bit_cast(f16x8, bits);
    f32x16 d = {0};
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = d[0];
}

int main() {
    {
        float* o; CK(hipMalloc(&o, 4));
        hipLaunchKernelGGL(denorm_check, dim3(1), dim3(64), 0, 0, o);
        float h; CK(hipMemcpy(&h, o, 4, hipMemcpyDeviceToHost));
        printf("f16 subnormal through MFMA: got %.9g, exact %.9g (%s)\n", h, 48.0 / 16777216.0, h == (float)(48.0 / 16777216.0) ? "kept" : "FLUSHED");
    }
    unsigned* in; CK(hipMalloc(&in, 4096)); CK(hipMemset(in, 0x5a, 4096));
    float* out; CK(hipMalloc(&out, 256 * 1024 * 4));
    long long* cyc; CK(hipMalloc(&cyc, 256 * 16 * 8));
    for (int wps = 1; wps <= 4; ++wps) {
        run<0>("dequant+mfma", wps, in, out, cyc);
        run<1>("dequant only", wps, in, out, cyc);
        run<2>("mfma only", wps, in, out, cyc);
        run<3>("deq+lds+mfma", wps, in, out, cyc);
        run<4>("deq+ldsAhead+mfma", wps, in, out, cyc);
        run<5>("subnormal+fold", wps, in, out, cyc);
    }
    return 0;
}
