// Calibration: cost of a grid-wide barrier inside one persistent launch on MI355X (8 XCDs), the price a fused
// "norm -> GEMM -> GEMM" launch would pay at every all-to-all seam instead of a kernel boundary (1.6 us + dispatch).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// counter barrier: one lane per block arrives with an agent-scope atomic, then polls
__global__ void barrier_kernel(unsigned* counter, int iters, int nblocks, unsigned* sink) {
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * nblocks;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        acc += it;
    }
    if (acc == 0x7fffffffu) *sink = acc;
}

// same with a payload: every block writes 4 KiB before the barrier and reads another block's 4 KiB after it
__global__ void barrier_payload_kernel(unsigned* counter, float* buf, int iters, int nblocks, float* sink) {
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float* mine = buf + ((size_t)(it & 1) * nblocks + blockIdx.x) * 1024;
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) mine[i] = (float)(it + i);
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * nblocks;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        const float* other = buf + ((size_t)(it & 1) * nblocks + (blockIdx.x + 37) % nblocks) * 1024;
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) acc += __builtin_nontemporal_load(other + i);
    }
    if (acc == 12345.678f) *sink = acc;
}

int main() {
    unsigned* counter; CK(hipMalloc(&counter, 4));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    float* buf; CK(hipMalloc(&buf, 2 * 1024 * 4096 * sizeof(float)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int nblocks : {64, 256, 512}) {
        for (int threads : {256, 1024}) {
            if (nblocks * threads > 256 * 2048) continue;
            const int iters = 2000;
            for (int mode = 0; mode < 2; ++mode) {
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(counter, 0, 4));
                    CK(hipEventRecord(a));
                    if (mode == 0)
                        hipLaunchKernelGGL(barrier_kernel, dim3(nblocks), dim3(threads), 0, 0, counter, iters, nblocks, sink);
                    else
                        hipLaunchKernelGGL(barrier_payload_kernel, dim3(nblocks), dim3(threads), 0, 0, counter, buf, iters, nblocks, (float*)sink);
                    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
                    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
                }
                printf("grid barrier%s blocks=%3d threads=%4d: %.2f us per barrier\n", mode ? " + 4 KiB exchange" : "", nblocks, threads, best * 1000.f / iters);
            }
        }
    }
    return 0;
}
