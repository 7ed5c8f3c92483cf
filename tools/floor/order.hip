// Does an L2-hit load issued behind a deep queue of HBM-miss loads wait for them?  (a) same wave, (b) another wave
// of the same CU that has no misses outstanding.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PER>
__global__ __launch_bounds__(1024) void order_kernel(const u32x4* __restrict__ src, const u32x4* __restrict__ hot,
                                                     long long* __restrict__ tout, unsigned* sink) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 16 + w;
    u32x4 v[PER];
    u32x4 h;
    long long t0 = __builtin_amdgcn_s_memtime();
    long long t_hot;
    if (w < 15) {
        const long base = wave * PER * 64 + lane;
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = __builtin_nontemporal_load(src + base + (long)i * 64);
        h = hot[lane + 64 * (w & 3)];                       // L2-resident, issued LAST
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_hot = __builtin_amdgcn_s_memtime();
    } else {
        h = hot[lane];                                       // wave 15: only the hot load
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_hot = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = h;
    }
    unsigned acc = h[0];
#pragma unroll
    for (int i = 0; i < PER; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    if (acc == 0x12345678u) *sink = acc;
    if (lane == 0) tout[wave] = t_hot - t0;
}

// variant: hot load issued FIRST by the streaming waves, time to vmcnt(PER)
template <int PER>
__global__ __launch_bounds__(1024) void first_kernel(const u32x4* __restrict__ src, const u32x4* __restrict__ hot,
                                                     long long* __restrict__ tout, unsigned* sink) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 16 + w;
    u32x4 v[PER];
    long long t0 = __builtin_amdgcn_s_memtime();
    u32x4 h = hot[lane + 64 * (w & 3)];
    const long base = wave * PER * 64 + lane;
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = __builtin_nontemporal_load(src + base + (long)i * 64);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    long long t_hot = __builtin_amdgcn_s_memtime();
    unsigned acc = h[0];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t_all = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < PER; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    if (acc == 0x12345678u) *sink = acc;
    if (lane == 0) { tout[wave * 2] = t_hot - t0; tout[wave * 2 + 1] = t_all - t0; }
}

static void stats(const char* name, std::vector<long long> v) {
    std::sort(v.begin(), v.end());
    printf("%-40s n=%zu  p10=%lld p50=%lld p90=%lld max=%lld (s_memtime ticks)\n", name, v.size(), v[v.size() / 10],
           v[v.size() / 2], v[v.size() * 9 / 10], v.back());
}

int main() {
    const int blocks = 172, PER = 16;
    const long bytes = (long)blocks * 16 * PER * 1024;
    unsigned char* buf; CK(hipMalloc(&buf, 2 * bytes)); CK(hipMemset(buf, 1, 2 * bytes));
    u32x4* hot; CK(hipMalloc(&hot, 1 << 16)); CK(hipMemset(hot, 2, 1 << 16));
    long long* tout; CK(hipMalloc(&tout, blocks * 16 * 2 * 8));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    std::vector<long long> h(blocks * 16 * 2);
    // warm the hot buffer into L2 with a first pass, then measure a second pass on fresh weights
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(order_kernel<PER>, dim3(blocks), dim3(1024), 0, 0, (const u32x4*)(buf + pass * bytes), hot, tout, sink);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), tout, blocks * 16 * 8, hipMemcpyDeviceToHost));
    std::vector<long long> same, other;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 16; ++w) (w < 15 ? same : other).push_back(h[b * 16 + w]);
    stats("hot load LAST, same wave (all done)", same);
    stats("hot load, idle wave of the same CU", other);
    CK(hipMemset(buf, 3, 2 * bytes));
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(first_kernel<PER>, dim3(blocks), dim3(1024), 0, 0, (const u32x4*)(buf + pass * bytes), hot, tout, sink);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), tout, blocks * 16 * 2 * 8, hipMemcpyDeviceToHost));
    std::vector<long long> first, all;
    for (int i = 0; i < blocks * 16; ++i) { first.push_back(h[2 * i]); all.push_back(h[2 * i + 1]); }
    stats("hot load FIRST: time to hot data", first);
    stats("hot load FIRST: time to all data", all);
    return 0;
}
