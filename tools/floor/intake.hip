// What can ONE CU take in?  (round 3: kernels with few blocks — the MQA split merge, 32-row norms, unsplit shard GEMMs —
// all sat at ~36 GB/s per block.)  `blocks` workgroups (one per CU) of W waves stream `per_block` KiB each from HBM with
// U 1-KiB loads in flight per wave (16 bytes per lane, non-temporal or plain); reports GB/s per CU and chip-wide.
//   hipcc --offload-arch=gfx950 -O3 -o tools/floor/intake tools/floor/intake.hip && tools/floor/intake
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ void stream(const u32x4* __restrict__ src, long per_block_kib, unsigned* sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const u32x4* p = src + (long)blockIdx.x * per_block_kib * 64 + lane;  // 64 u32x4 per KiB
    u32x4 acc = {0, 0, 0, 0};
    for (long i = w * U; i + U <= per_block_kib; i += (long)nw * U) {
        u32x4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = NT ? __builtin_nontemporal_load(p + (i + j) * 64) : p[(i + j) * 64];
#pragma unroll
        for (int j = 0; j < U; ++j) acc |= v[j];
    }
    if ((acc[0] & acc[1] & acc[2] & acc[3]) == 0x12345677u) *sink = 1;
}

// the same stream in scattered 8-KiB pieces (a KV page of one head: 32 tokens x 128 x 2 bytes), U KiB in flight per wave
template <int PK>
__global__ void stream_pages(const u32x4* __restrict__ src, long per_block_kib, long total_pages, unsigned* sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    const long npages = per_block_kib / PK;
    for (long pg = w; pg < npages; pg += nw) {
        // a pseudo-random page of the 3 GiB buffer (distinct per (block, pg) with overwhelming probability)
        const unsigned long h = ((unsigned long)blockIdx.x * 1000003ul + (unsigned long)pg) * 0x9E3779B97F4A7C15ul;
        const u32x4* p = src + (long)((h >> 20) % (unsigned long)total_pages) * (64 * PK) + lane;
#pragma unroll
        for (int j0 = 0; j0 < PK; j0 += 8) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + (j0 + j) * 64);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc |= v[j];
        }
    }
    if ((acc[0] & acc[1] & acc[2] & acc[3]) == 0x12345677u) *sink = 1;
}

template <int PK>
void run_pages(const u32x4* buf, long total_kib, unsigned* sink, int blocks, int waves, long per_block_kib) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((stream_pages<PK>), dim3(blocks), dim3(64 * waves), 0, 0, buf, per_block_kib, total_kib / PK, sink);
        CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r > 0 && ms < best) best = ms;
    }
    const double us = best * 1e3 - 2.0;
    printf("scattered %2d-KiB pieces: blocks %4d waves %2d: %7.1f us for %5ld KiB per block -> %6.1f GB/s per block, %6.0f GB/s chip\n", PK, blocks, waves,
           best * 1e3, per_block_kib, per_block_kib * 1024.0 / us / 1e3, per_block_kib * 1024.0 * blocks / us / 1e3);
}

template <int U, int NT>
void run(const u32x4* buf, long total_kib, unsigned* sink, int blocks, int waves, long per_block_kib) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    const long sets = total_kib / (per_block_kib * blocks);
    for (int r = 0; r < 6; ++r) {
        const u32x4* s = buf + (long)(r % sets) * per_block_kib * blocks * 64;
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((stream<U, NT>), dim3(blocks), dim3(64 * waves), 0, 0, s, per_block_kib, sink);
        CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r > 0 && ms < best) best = ms;
    }
    const double us = best * 1e3 - 2.0;  // ~2 us of launch + ramp are not intake
    printf("blocks %3d waves %2d in flight/wave %2d KiB %s: %7.1f us for %5ld KiB per block -> %6.1f GB/s per CU, %6.0f GB/s chip\n", blocks, waves, U,
           NT ? "nt" : "  ", best * 1e3, per_block_kib, per_block_kib * 1024.0 / us / 1e3, per_block_kib * 1024.0 * blocks / us / 1e3);
}

int main() {
    const long total_kib = 3L << 20;  // 3 GiB
    u32x4* buf; CK(hipMalloc(&buf, total_kib * 1024));
    CK(hipMemset(buf, 1, total_kib * 1024));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    for (int waves : {2, 4}) {   // the cfg3 attention geometry: 1024 blocks x 512 KiB, K and V pages of 8 KiB each
        run_pages<8>(buf, total_kib, sink, 1024, waves, 512);
        run_pages<16>(buf, total_kib, sink, 1024, waves, 512);
        run_pages<32>(buf, total_kib, sink, 1024, waves, 512);
        run_pages<64>(buf, total_kib, sink, 1024, waves, 512);
    }
    for (int blocks : {1, 32, 128, 256}) {
        const long kib = blocks == 256 ? 1024 : 2048;
        for (int waves : {4, 8, 16}) {
            run<2, 1>(buf, total_kib, sink, blocks, waves, kib);
            run<4, 1>(buf, total_kib, sink, blocks, waves, kib);
            run<8, 1>(buf, total_kib, sink, blocks, waves, kib);
            run<16, 1>(buf, total_kib, sink, blocks, waves, kib);
        }
        run<8, 0>(buf, total_kib, sink, blocks, 8, kib);
    }
    return 0;
}
