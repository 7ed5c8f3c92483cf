// Round 4: the PRODUCT's fragment-order int4 GEMM unit (csrc/gptq_wide_body.h, included as is) on arbitrary shapes, with
// per-wave s_memtime stamps: where does a launch of the 64-row form (MR = 2) spend its time at the Llama-2-70B shapes?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWIDE_TRACE] -I text-generation-inference_amd/csrc -o tools/floor/wide_unit tools/floor/wide_unit.hip
//   tools/floor/wide_unit            (random images: timing and timeline only; parity is tests/test_fragments_gpu.py's job)
#include <hip/hip_runtime.h>
// Two builds: plain (timing: the unit exactly as the library compiles it) and -DWIDE_TRACE (per-wave stamps).  The stamps must
// not be in the timed build: a store inside the loop makes hipcc give up its counted vmcnt waits (loads and stores complete
// out of order with respect to each other), i.e. the traced kernel drains to vmcnt(0) once per iteration — its timeline shows
// where the time goes, its absolute numbers are a few percent above the library's.
#ifdef WIDE_TRACE
static __device__ long long* g_wide_trace = nullptr;   // [blocks][8 waves][8]
#define WIDE_STAMP(i)                                                                                                    \
    do {                                                                                                                 \
        if (g_wide_trace) {                                                                                              \
            const long long t_ = __builtin_amdgcn_s_memtime();                                                           \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
            if ((threadIdx.x & 63) == 0)                                                                                 \
                g_wide_trace[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 8 + (i)] = t_;  \
        }                                                                                                                \
    } while (0)
#endif
#include "gptq_wide_body.h"
#include <vector>
#include <algorithm>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int CT, int ACT, bool OUTF, int MR>
__global__ __launch_bounds__(64 * gptq::WIDE_WK) void unit_kernel(gptq::GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gptq::gptq_wide_unit<CT, ACT, OUTF, MR>(a, smem);
}

static uint64_t rng_state = 0x1234567ull;
static inline uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }

struct Image { int K, N, NT, KS, G; int64_t offB, total; };
static Image layout(int K, int N, int gs) {
    Image im; im.K = K; im.N = N; im.G = K / gs; im.NT = (N + 31) / 32; im.KS = ((K + 255) / 256) * 4 + 1;
    im.offB = (int64_t)im.NT * im.KS * 1024;
    im.total = (im.offB + (int64_t)im.NT * im.G * 128 + 255) & ~255ll;
    return im;
}

template <int CT, int ACT, int MR>
static void run(const char* name, int K, int N, int S, int M) {
    const Image im = layout(K, N, 128);
    const int nsets = (int)std::max<int64_t>(2, (900ll << 20) / im.total);
    std::vector<uint8_t> host(im.total);
    uint32_t* w = reinterpret_cast<uint32_t*>(host.data());
    for (int64_t i = 0; i < im.offB / 4; ++i) w[i] = rnd() ^ (rnd() << 16);
    uint32_t* sz = reinterpret_cast<uint32_t*>(host.data() + im.offB);
    for (int64_t i = 0; i < (int64_t)im.NT * im.G * 32; ++i) {
        _Float16 s = (_Float16)(0.005f + 0.01f * (rnd() % 1000) / 1000.f), z = (_Float16)(1024.f + (rnd() % 14) + 1);
        uint16_t sb, zb; memcpy(&sb, &s, 2); memcpy(&zb, &z, 2);
        sz[i] = sb | ((uint32_t)zb << 16);
    }
    std::vector<uint8_t*> sets(nsets);
    for (auto& p : sets) { CK(hipMalloc(&p, im.total)); CK(hipMemcpy(p, host.data(), im.total, hipMemcpyHostToDevice)); }
    std::vector<_Float16> hx((size_t)32 * MR * K);
    for (auto& v : hx) v = (_Float16)(((int)(rnd() % 2001) - 1000) / 1000.f);
    _Float16* dx; CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    _Float16* dout; CK(hipMalloc(&dout, (size_t)32 * MR * N * 2));
    float* dslabs; CK(hipMalloc(&dslabs, (size_t)MR * 16 * 32 * im.NT * 32 * 4));
    gptq::GemmArgs a; memset(&a, 0, sizeof(a));
    a.x = dx; a.ldx = 0; a.offB = im.offB; a.out = dout; a.ldo = ACT == 2 ? N / 2 : N; a.M = M; a.K = K; a.N = N;
    a.G = im.G; a.gs = 128; a.S = S; a.NT = im.NT; a.KS = im.KS; a.slabs = dslabs; a.partial = S > 1;
    a.spg_shift = 1;
    const int cgs = (im.NT + CT - 1) / CT;
    const size_t lds = gptq::wide_lds_bytes(CT);
    auto kern = unit_kernel<CT, ACT, false, MR>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(cgs, S);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    const int iters = 2 * nsets;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) { a.prep = sets[i % nsets]; hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, a); }
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 1000.f / iters);
    }
    const double mb = ((double)K * N / 2 + (double)im.NT * im.G * 128) / 1e6;
    printf("%-22s M %2d CT %d S %d act %d: blocks %4d  %7.2f us  %5.2f TB/s\n", name, M, CT, S, ACT, cgs * S, best, mb / best);
#ifdef WIDE_TRACE
    // timeline of one cold launch
    const int nw = cgs * S * 8;
    long long* dtr; CK(hipMalloc(&dtr, (size_t)nw * 64)); CK(hipMemset(dtr, 0, (size_t)nw * 64));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_wide_trace), &dtr, sizeof(dtr)));
    for (int i = 0; i < 3; ++i) { a.prep = sets[(i + 1) % nsets]; hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, a); }
    CK(hipDeviceSynchronize());
    long long* nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_wide_trace), &nul, sizeof(nul)));
    std::vector<long long> tr((size_t)nw * 8);
    CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
    // s_memtime counters differ between XCDs: every wave's stamps are taken relative to its own entry
    const char* names[5] = {"entry", "prologue issued", "loop done", "last steps done", "stored"};
    printf("    ticks after the wave's own entry (min / median / max over %d waves):\n", nw);
    for (int k = 1; k < 5; ++k) {
        std::vector<long long> v(nw);
        for (int i = 0; i < nw; ++i) v[i] = tr[(size_t)i * 8 + k] - tr[(size_t)i * 8];
        std::sort(v.begin(), v.end());
        printf("      %-18s %8lld %8lld %8lld\n", names[k], v[0], v[nw / 2], v[nw - 1]);
    }
    CK(hipFree(dtr));
#endif
    for (auto p : sets) CK(hipFree(p));
    CK(hipFree(dx)); CK(hipFree(dout)); CK(hipFree(dslabs));
}

int main() {
    run<4, 2, 2>("70B gate_up", 8192, 57344, 1, 64);
    run<7, 2, 2>("70B gate_up", 8192, 57344, 1, 64);
    run<8, 2, 2>("70B gate_up", 8192, 57344, 1, 64);
    run<2, 0, 2>("70B o", 8192, 8192, 2, 64);
    run<4, 0, 2>("70B o", 8192, 8192, 4, 64);
    run<2, 0, 2>("70B down", 28672, 8192, 2, 64);
    run<4, 0, 2>("70B down", 28672, 8192, 4, 64);
    run<8, 0, 2>("70B down", 28672, 8192, 8, 64);
    run<3, 2, 2>("7B gate_up", 4096, 22016, 1, 64);
    run<2, 0, 2>("7B down", 11008, 4096, 4, 64);
    run<4, 0, 2>("7B down", 11008, 4096, 8, 64);
    return 0;
}
