// Experiment (round 3): pull a byte range into the memory-side cache (Infinity Cache, 256 MiB) and the L2s ahead of the
// kernel that will stream it.  `blocks` workgroups of 256 threads sweep the range with 16-byte loads whose results are
// dropped; the launch is on the caller's stream, so it can sit in a captured graph in front of / beside a GEMM.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/floor/libprefetch.so tools/floor/prefetch.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void prefetch_kernel(const unsigned char* __restrict__ p, long n16, unsigned* sink) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    // 8 loads in flight per thread
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x4* q = reinterpret_cast<const u32x4*>(p) + i + j * stride;
            v[j] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc |= v[j];
    }
    for (; i < n16; i += stride) acc |= reinterpret_cast<const u32x4*>(p)[i];
    if ((acc[0] & acc[1] & acc[2] & acc[3]) == 0x12345677u && sink) *sink = 1;  // keeps the loads alive
}

extern "C" int prefetch(const void* p, long bytes, int blocks, int nt, void* stream) {
    if (nt)
        hipLaunchKernelGGL(prefetch_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)p, bytes / 16, nullptr);
    else
        hipLaunchKernelGGL(prefetch_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)p, bytes / 16, nullptr);
    return (int)hipGetLastError();
}
