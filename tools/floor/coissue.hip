// Do VALU and MFMA instructions of a SIMD overlap?  (round 3, after the 64-row int4 GEMM loop measured VALU + MFMA, not
// max(VALU, MFMA), per k64-step.)  Per iteration: 8 independent v_mfma_f32_32x32x16_f16 (4 accumulators), and / or 56
// packed-f16 / integer VALU instructions, either as two blocks or interleaved 7 per MFMA; 1, 2, 3 or 4 waves per SIMD.
// The VALU registers are never MFMA operands (no data hazards between the two streams).
//   hipcc --offload-arch=gfx950 -O3 -o tools/floor/coissue tools/floor/coissue.hip && tools/floor/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// 7 VALU: the dequantisation mix (and_or, pk_add, pk_fma)
#define VALU7(i) asm volatile( \
    "v_and_or_b32 %0, %0, %8, %9\n\tv_pk_fma_f16 %1, %1, %10, %11\n\tv_pk_add_f16 %2, %2, %10\n\tv_and_or_b32 %3, %3, %8, %9\n\t" \
    "v_pk_fma_f16 %4, %4, %10, %11\n\tv_pk_add_f16 %5, %5, %10\n\tv_pk_fma_f16 %6, %6, %10, %11\n\tv_lshrrev_b32 %7, 4, %7" \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
    : "v"(m0), "v"(m1), "v"(c0), "v"(c1))

template <int MODE>  // 0 MFMA only, 1 VALU only, 2 blocks (56 VALU, then 8 MFMA), 3 interleaved
__global__ __launch_bounds__(1024) void k(unsigned* out, long long* cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    unsigned r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i;
    unsigned m0 = 0x000f000f, m1 = 0x64006400, c0 = 0x3c003c00, c1 = 0x00000000;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
        if (MODE == 1) { VALU7(0); VALU7(1); VALU7(2); VALU7(3); VALU7(4); VALU7(5); VALU7(6); VALU7(7); }
        if (MODE == 2) {
            VALU7(0); VALU7(1); VALU7(2); VALU7(3); VALU7(4); VALU7(5); VALU7(6); VALU7(7);
            MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3);
        }
        if (MODE == 3) {
            MFMA(acc0); VALU7(0); MFMA(acc1); VALU7(1); MFMA(acc2); VALU7(2); MFMA(acc3); VALU7(3);
            MFMA(acc0); VALU7(4); MFMA(acc1); VALU7(5); MFMA(acc2); VALU7(6); MFMA(acc3); VALU7(7);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    float f = 0;
    for (int i = 0; i < 16; ++i) f += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    if (s == 0x12345 && f == 1.25f) out[0] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned* out, long long* cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 4; ++wps) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        // wave 0 is the oldest wave of its SIMD and is served first: the kernel time is what counts
        printf("%-28s waves/SIMD=%d: kernel %7.1f ns per iteration-round = %6.1f ns per wave-iteration per SIMD   (wave 0: %.1f ticks per iteration)\n",
               name, wps, ms * 1e6 / iters, ms * 1e6 / iters / wps, (double)c / iters);
    }
}

int main() {
    unsigned* out; long long* cyc;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&cyc, 8));
    run<0>("8 MFMA 32x32x16", out, cyc);
    run<1>("56 VALU", out, cyc);
    run<2>("56 VALU, then 8 MFMA", out, cyc);
    run<3>("8 x (MFMA, 7 VALU)", out, cyc);
    return 0;
}
