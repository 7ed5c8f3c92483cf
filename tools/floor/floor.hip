// Calibration: per-launch cost of (a) an empty kernel, (b) a pure streaming read of N MB, inside a HIP graph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }

// each wave reads PER consecutive KiB-blocks (all loads issued before the first use)
template <int PER>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ src, long n16, unsigned* sink) {
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    u32x4 v[PER];
    const long base = wave * PER * 64 + lane;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        long idx = base + (long)i * 64;
        if (idx >= n16) idx = lane;
        v[i] = __builtin_nontemporal_load(src + idx);
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    if (acc == 0x12345678u) *sink = acc;
}

template <typename F> float time_graph(F launch, int reps, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best * 1000.f / reps;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const long total = 3L << 30;  // rotate through 3 GiB so nothing stays cached
    unsigned char* buf; CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    printf("empty kernel 1x64: %.2f us/launch\n", time_graph([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, nullptr); }, 200, st));
    printf("empty kernel 256x1024: %.2f us/launch\n", time_graph([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(1024), 0, st, nullptr); }, 200, st));
    printf("empty kernel 1024x256: %.2f us/launch\n", time_graph([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, st, nullptr); }, 200, st));
    const double sizes_mb[] = {8.4, 22.5, 25.2, 45.1, 128.0};
    for (double mb : sizes_mb) {
        long bytes = (long)(mb * 1e6) / 1024 * 1024;
        long n16 = bytes / 16;
        long slots = total / bytes;
        auto run = [&](auto kern, int per, int threads, const char* name) {
            long waves = (n16 / 64 + per - 1) / per;
            long blocks = (waves * 64 + threads - 1) / threads;
            float us = time_graph([&](int i) {
                const u32x4* s = (const u32x4*)(buf + (long)(i % slots) * bytes);
                hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), 0, st, s, n16, sink);
            }, 60, st);
            printf("stream %6.1f MB %-22s blocks=%5ld: %6.2f us  %6.0f GB/s\n", mb, name, blocks, us, bytes / us / 1e3);
        };
        run(stream_kernel<4>, 4, 256, "per=4 t=256");
        run(stream_kernel<4>, 4, 1024, "per=4 t=1024");
        run(stream_kernel<8>, 8, 256, "per=8 t=256");
        run(stream_kernel<8>, 8, 1024, "per=8 t=1024");
        run(stream_kernel<16>, 16, 256, "per=16 t=256");
        run(stream_kernel<16>, 16, 1024, "per=16 t=1024");
        run(stream_kernel<32>, 32, 512, "per=32 t=512");
    }
    return 0;
}
