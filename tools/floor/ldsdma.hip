// Calibration of the LDS-DMA weight stream (round 3): one loader wave per CU streams its share of N MB into an LDS ring
// with `global_load_lds_dwordx4` (1 KiB per instruction), throttled by a counted vmcnt.  Reports the chip rate for
// different in-flight caps, cache policies and ring placements (below / above 64 KiB of LDS), and checks that what
// landed in LDS is what was in memory (the last ring fill is compared with plain loads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ void dma1k(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// grid = CUs; block = 64 * (1 + EXTRA) threads: wave 0 loads, the others (if any) idle at the barrier.
// Each block streams `per_block` KiB starting at src + block * per_block KiB into a ring of SLOTS KiB at LDS offset BASE.
// L2 mode: all blocks sweep the SAME `win_kib` KiB window (L2 / MALL resident) `per_block` KiB long
template <int VMAX, int MIX>
__global__ __launch_bounds__(64) void dma_l2(const unsigned char* __restrict__ win, long win_kib, const unsigned char* __restrict__ hbm,
                                              long per_block, long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const unsigned ring = (unsigned)(size_t)(smem);
    const unsigned char* mine = hbm + (long)blockIdx.x * per_block * 1024 + lane * 16;
    long long t0 = __builtin_amdgcn_s_memtime();
    int slot = 0;
    long wpos = (blockIdx.x * 7) % win_kib;
    for (long i = 0; i < per_block; ++i) {
        dma1k<0>(win + wpos * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(ring + slot * 1024));
        wait_vm<VMAX>();
        slot = (slot + 1) & 31;
        wpos = wpos + 1 == win_kib ? 0 : wpos + 1;
        if (MIX) {
            dma1k<1>(mine + i * 1024, __builtin_amdgcn_readfirstlane(ring + slot * 1024));
            wait_vm<VMAX>();
            slot = (slot + 1) & 31;
        }
    }
    wait_vm<0>();
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SLOTS, int VMAX, int NT, int BASE>
__global__ __launch_bounds__(256) void dma_stream(const unsigned char* __restrict__ src, long per_block, unsigned* __restrict__ bad,
                                                   long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned ring = (unsigned)(size_t)(smem) + BASE;  // LDS byte address of the ring
    const unsigned char* mine = src + (long)blockIdx.x * per_block * 1024 + lane * 16;
    long long t0 = 0, t1 = 0;
    if (w == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        int slot = 0;
        for (long i = 0; i < per_block; ++i) {
            dma1k<NT>(mine + i * 1024, __builtin_amdgcn_readfirstlane(ring + slot * 1024));
            wait_vm<VMAX>();
            slot = slot + 1 == SLOTS ? 0 : slot + 1;
        }
        wait_vm<0>();
        t1 = __builtin_amdgcn_s_memtime();
    }
    __syncthreads();
    // check the last SLOTS KiB (or fewer) against memory
    if (w == 0) {
        const long n = per_block < SLOTS ? per_block : SLOTS;
        unsigned wrong = 0;
        for (long j = 0; j < n; ++j) {
            const long i = per_block - n + j;
            const int slot = (int)(i % SLOTS);
            u32x4 got = *reinterpret_cast<const u32x4*>(smem + BASE + slot * 1024 + lane * 16);
            u32x4 want = *reinterpret_cast<const u32x4*>(mine + i * 1024);
            wrong += (got[0] != want[0]) + (got[1] != want[1]) + (got[2] != want[2]) + (got[3] != want[3]);
        }
        if (wrong) atomicAdd(bad, wrong);
        if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    }
}

template <int SLOTS, int VMAX, int NT, int BASE>
void run(const char* name, const unsigned char* buf, long total_bytes, double mb, unsigned* bad, long long* cyc, hipStream_t st) {
    const int blocks = 256;
    long per_block = (long)(mb * 1e6 / 1024 / blocks);
    const long bytes = per_block * 1024 * blocks;
    const long sets = total_bytes / bytes;
    const size_t lds = BASE + SLOTS * 1024;
    CK(hipFuncSetAttribute((const void*)dma_stream<SLOTS, VMAX, NT, BASE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipMemsetAsync(bad, 0, 4, st));
    const int reps = 40;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((dma_stream<SLOTS, VMAX, NT, BASE>), dim3(blocks), dim3(64), lds, st, buf + (long)(i % sets) * bytes,
                           per_block, bad, cyc);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    long long hc[4]; CK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
    const float us = best * 1000.f / reps;
    printf("%-44s %6.1f MB: %6.2f us/launch  %6.0f GB/s   loader cycles %lld  mismatched words %u\n", name, bytes / 1e6, us,
           bytes / us / 1e3, hc[0], hb);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const long total = 3L << 30;
    unsigned char* buf; CK(hipMalloc(&buf, total));
    {  // non-trivial content
        unsigned* h = (unsigned*)malloc(64 << 20);
        for (long i = 0; i < (64 << 20) / 4; ++i) h[i] = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7);
        for (long off = 0; off < total; off += 64 << 20) CK(hipMemcpy(buf + off, h, 64 << 20, hipMemcpyHostToDevice));
        free(h);
    }
    unsigned* bad; CK(hipMalloc(&bad, 4));
    long long* cyc; CK(hipMalloc(&cyc, 256 * 8));
    {
        CK(hipFuncSetAttribute((const void*)dma_l2<24, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        CK(hipFuncSetAttribute((const void*)dma_l2<24, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        for (int blocks : {64, 256}) {
            const long per_block = 176;
            for (int mix = 0; mix < 2; ++mix) {
                hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(a, st));
                    if (mix) hipLaunchKernelGGL((dma_l2<24, 1>), dim3(blocks), dim3(64), 32 * 1024, st, buf, 256L, buf + (1L << 30) + (long)rep * (64 << 20), per_block, cyc);
                    else hipLaunchKernelGGL((dma_l2<24, 0>), dim3(blocks), dim3(64), 32 * 1024, st, buf, 256L, buf + (1L << 30), per_block, cyc);
                    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
                }
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                long long hc[4]; CK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
                printf("L2-window DMA %s: %d blocks x %ld KiB: %.2f us, loader cycles %lld -> %.1f B/clk/CU for the %s\n", mix ? "+ HBM stream 1:1" : "only", blocks, per_block, ms * 1e3, hc[0],
                       (mix ? 2.0 : 1.0) * per_block * 1024.0 / hc[0], mix ? "sum" : "window");
            }
        }
    }
    const double sizes[] = {45.1};
    for (double mb : sizes) {
        run<16, 8, 0, 0>("ring 16K vmcnt 8", buf, total, mb, bad, cyc, st);
        run<32, 16, 0, 0>("ring 32K vmcnt 16", buf, total, mb, bad, cyc, st);
        run<48, 32, 0, 0>("ring 48K vmcnt 32", buf, total, mb, bad, cyc, st);
        run<64, 48, 0, 0>("ring 64K vmcnt 48", buf, total, mb, bad, cyc, st);
        run<64, 48, 1, 0>("ring 64K vmcnt 48 nt", buf, total, mb, bad, cyc, st);
        run<64, 60, 1, 0>("ring 64K vmcnt 60 nt", buf, total, mb, bad, cyc, st);
        run<64, 48, 1, 65536>("ring 64K at LDS+64K vmcnt 48 nt", buf, total, mb, bad, cyc, st);
        run<96, 60, 1, 32768>("ring 96K at LDS+32K vmcnt 60 nt", buf, total, mb, bad, cyc, st);
    }
    return 0;
}
