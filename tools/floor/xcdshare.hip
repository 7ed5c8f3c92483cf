// Round 5: the decode attention's waves on XCD 3 and 5 finish ~8 % after those on XCD 0 / 6 / 7, whatever heads they read
// (profiles/r05_attn_rot.log).  Is that the XCDs' own intake (then moving work to the others would shorten the launch), or their
// share of a saturated HBM (then it would not)?  A load-only stream in the attention's geometry (1024 blocks x 2 waves, 512 MiB
// per launch, block b on XCD b % 8) with the bytes dealt evenly, then re-dealt in proportion to each XCD's measured rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/floor/xcdshare tools/floor/xcdshare.hip && tools/floor/xcdshare
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Share { int chunks[8]; int first[8]; };  // 16 KiB chunks per block of XCD x, first chunk of XCD x's region (per block row)

// block b = XCD (b % 8), row (b / 8): reads chunks [first[x] + row * chunks[x], + chunks[x]) of 16 KiB, its two waves alternating
__global__ void stream(const u32x4* __restrict__ src, Share sh, long long* done, unsigned* sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int x = blockIdx.x & 7, row = blockIdx.x >> 3;
    const long c0 = (long)sh.first[x] + (long)row * sh.chunks[x];
    u32x4 acc = {0, 0, 0, 0};
    for (int i = w; i < sh.chunks[x]; i += 2) {
        const u32x4* p = src + (c0 + i) * 1024 + lane;
        u32x4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc |= v[j];
    }
    if ((acc[0] & acc[1] & acc[2] & acc[3]) == 0x12345677u) *sink = 1;
    if (lane == 0) done[blockIdx.x * 2 + w] = __builtin_amdgcn_s_memrealtime();
}

int main() {
    const long total_chunks = 32768;  // x 16 KiB = 512 MiB
    const int sets = 4, blocks = 1024;
    u32x4* bufs[sets];
    for (auto& b : bufs) { CK(hipMalloc(&b, (total_chunks + 4096) * 16384)); CK(hipMemset(b, 1, (total_chunks + 4096) * 16384)); }
    unsigned* sink; CK(hipMalloc(&sink, 4));
    long long* done; CK(hipMalloc(&done, blocks * 2 * 8));
    std::vector<long long> h(blocks * 2);
    double w[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int iter = 0; iter < 5; ++iter) {
        Share sh; double ws = 0; for (double v : w) ws += v;
        int first = 0;
        for (int x = 0; x < 8; ++x) {
            sh.chunks[x] = std::max(2, (int)(total_chunks / 128.0 * w[x] / ws + 0.5));   // per block (128 blocks per XCD)
            sh.first[x] = first; first += sh.chunks[x] * 128;
        }
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream, dim3(blocks), dim3(128), 0, 0, bufs[i % sets], sh, done, sink);
        CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(stream, dim3(blocks), dim3(128), 0, 0, bufs[i % sets], sh, done, sink);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms * 1000.f / 8);
        }
        // five more launches for the finish times (100 MHz clock; medians per XCD relative to the launch's last finish, averaged)
        double fin[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long t0 = 0, t1 = 0;
        for (int m = 0; m < 5; ++m) {
            hipLaunchKernelGGL(stream, dim3(blocks), dim3(128), 0, 0, bufs[m % sets], sh, done, sink);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), done, h.size() * 8, hipMemcpyDeviceToHost));
            t0 = *std::min_element(h.begin(), h.end()); t1 = *std::max_element(h.begin(), h.end());
            for (int x = 0; x < 8; ++x) {
                std::vector<long long> v;
                for (int b = x; b < blocks; b += 8) { v.push_back(h[2 * b]); v.push_back(h[2 * b + 1]); }
                std::sort(v.begin(), v.end());
                fin[x] += (v[v.size() / 2] - (t1 - (long long)(best * 100))) / 100.0 / 5;   // us after the (estimated) start of the launch
            }
        }
        const double mb = (double)first * 16384 / 1e6;
        printf("deal %d: chunks per block by XCD", iter);
        for (int x = 0; x < 8; ++x) printf(" %d", sh.chunks[x]);
        printf("  (%.0f MB)  %.2f us per launch = %.2f TB/s;  median finish by XCD (us):", mb, best, mb / best);
        for (int x = 0; x < 8; ++x) printf(" %.1f", fin[x]);
        printf("   last - first finish %.1f us\n", (t1 - t0) / 100.0);
        // re-deal: an XCD that finished late gets less (its share x mean finish / its finish)
        double mean = 0; for (double f : fin) mean += f / 8;
        for (int x = 0; x < 8; ++x) w[x] *= mean / std::max(fin[x], 1.0);
    }
    return 0;
}
