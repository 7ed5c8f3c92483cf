// Calibration (round 2): grid-wide barrier inside one persistent launch, re-measured with the forms
// MI355X_MICROARCH.md prices ("barrier-counter" 7.4 us, "barrier-xcd" 4.1-4.7 us at 256 workgroups) instead of the
// acquire-polling flat counter of gridbar.hip (10.9 us).  Variants, all with relaxed agent-scope polling + s_sleep:
//   flat      one monotonic counter, no fences (inter-block data would travel as sc1 write-through stores / sc1 loads)
//   two-level 8 group counters (group = block % 8) -> top counter -> 8 generation words, no fences
//   xcd       groups = the PHYSICAL XCC of each block (HW_REG_XCC_ID), group leader = last arriver does ONE
//             release fence (buffer_wbl2 of its XCD's L2) before the top counter, every block ONE acquire fence after
//   +payload  every block publishes 4 KiB (sc1 stores) before and reads another block's 4 KiB (sc1 loads) after
// Every spin is bounded (a stuck barrier reports instead of hanging the box).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RLX_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RLX_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
constexpr unsigned SPIN_LIMIT = 4000000;  // ~1 s

struct Bar {
    unsigned flat;            unsigned pad0[31];
    unsigned top;             unsigned pad1[31];
    unsigned grp[8][32];      // one 128-byte line per group counter
    unsigned gen[8][32];      // one line per generation word
    unsigned census[8][32];   // blocks per physical XCC (xcd variant)
    unsigned fail;
};

__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned target, unsigned* fail) {
    for (unsigned spins = 0; RLX_LOAD(p) < target; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (spins > SPIN_LIMIT) { RLX_STORE(fail, 1u); return false; }
    }
    return true;
}

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }  // HW_REG_XCC_ID[3:0]

// mode 0 flat, 1 two-level (logical groups), 2 xcd (physical groups + release/acquire fences)
template <int MODE>
__device__ __forceinline__ void grid_barrier(Bar* b, unsigned epoch, unsigned nblocks, unsigned my_group, unsigned group_size) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 0) {
            RLX_ADD(&b->flat, 1u);
            wait_ge(&b->flat, epoch * nblocks, &b->fail);
        } else {
            const unsigned old = RLX_ADD(&b->grp[my_group][0], 1u);
            if (old + 1 == epoch * group_size) {  // last arriver of the group
                if (MODE == 2) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                const unsigned t = RLX_ADD(&b->top, 1u);
                if (t + 1 == epoch * 8u) {
#pragma unroll
                    for (int g = 0; g < 8; ++g) RLX_STORE(&b->gen[g][0], epoch);
                }
            }
            wait_ge(&b->gen[my_group][0], epoch, &b->fail);
            if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}

template <int MODE, bool PAYLOAD>
__global__ void bar_kernel(Bar* b, float* buf, int iters, float* sink) {
    const unsigned nblocks = gridDim.x;
    unsigned group = blockIdx.x & 7u, gsize = (nblocks - group + 7) / 8;
    unsigned epoch = 0;
    if (MODE == 2) {  // census of the physical placement, then one flat barrier so that every block can read it
        group = xcc_id();
        if (threadIdx.x == 0) RLX_ADD(&b->census[group][0], 1u);
        grid_barrier<0>(b, ++epoch, nblocks, 0, 0);
        gsize = RLX_LOAD(&b->census[group][0]);
    }
    float acc = 0.f;
    unsigned e2 = 0;
    for (int it = 0; it < iters; ++it) {
        if (PAYLOAD) {
            float* mine = buf + ((size_t)(it & 1) * nblocks + blockIdx.x) * 1024;
            for (int i = threadIdx.x * 4; i < 1024; i += blockDim.x * 4) {
                float4 v = make_float4((float)(it + i), 1.f, 2.f, 3.f);
                if (MODE == 2) *reinterpret_cast<float4*>(mine + i) = v;  // plain stores: the leader's release fence publishes them
                else __builtin_nontemporal_store(v.x, mine + i), __builtin_nontemporal_store(v.y, mine + i + 1),
                     __builtin_nontemporal_store(v.z, mine + i + 2), __builtin_nontemporal_store(v.w, mine + i + 3);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        grid_barrier<MODE>(b, MODE == 0 ? ++epoch : ++e2, nblocks, group, gsize);
        if (PAYLOAD) {
            const float* other = buf + ((size_t)(it & 1) * nblocks + (blockIdx.x + 37) % nblocks) * 1024;
            for (int i = threadIdx.x; i < 1024; i += blockDim.x) acc += __builtin_nontemporal_load(other + i);
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

int main() {
    Bar* bar; CK(hipMalloc(&bar, sizeof(Bar)));
    float* sink; CK(hipMalloc(&sink, 4));
    float* buf; CK(hipMalloc(&buf, 2 * 1024 * 1024 * sizeof(float)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[3] = {"flat", "two-level", "xcd"};
    for (int nblocks : {256, 512}) {
        for (int threads : {256, 512, 1024}) {
            if (nblocks * threads > 256 * 2048) continue;
            const int iters = 1000;
            for (int mode = 0; mode < 3; ++mode) for (int payload = 0; payload < 2; ++payload) {
                float best = 1e9; unsigned fail = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(bar, 0, sizeof(Bar)));
                    CK(hipEventRecord(a));
#define L(M, P) hipLaunchKernelGGL((bar_kernel<M, P>), dim3(nblocks), dim3(threads), 0, 0, bar, buf, iters, sink)
                    if (mode == 0) { if (payload) L(0, true); else L(0, false); }
                    if (mode == 1) { if (payload) L(1, true); else L(1, false); }
                    if (mode == 2) { if (payload) L(2, true); else L(2, false); }
                    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
                    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
                    CK(hipMemcpy(&fail, &bar->fail, 4, hipMemcpyDeviceToHost));
                }
                printf("grid barrier %-9s%s blocks=%3d threads=%4d: %6.2f us per barrier%s\n", names[mode],
                       payload ? " +4KiB" : "      ", nblocks, threads, best * 1000.f / iters, fail ? "  (SPIN LIMIT HIT)" : "");
            }
        }
    }
    return 0;
}
