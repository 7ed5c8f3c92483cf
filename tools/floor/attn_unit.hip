// Round 5: the PRODUCT's decode attention (csrc/attention.hip, included as is) with per-wave s_memtime stamps — where does a
// launch spend its time outside the K/V stream?  (cfg3 MHA at 0.77 of the HBM peak, cfg5 MQA at 0.29.)
//   stamps (ticks from the wave's own entry): 1 sequence lengths in, 2 first block-table entry in (q requested),
//   3 first page applied, 4 all pages applied, 5 every wave's O in LDS (after the block barrier), 6 output / split record stored
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I text-generation-inference_amd/csrc -o tools/floor/attn_unit \
//         tools/floor/attn_unit.hip text-generation-inference_amd/csrc/attention_prefill.hip text-generation-inference_amd/csrc/api.hip
//   tools/floor/attn_unit            (cfg3 MHA B 32 ctx 1023 f16;  cfg5 MQA 48:1 B 32 ctx 4095 bf16;  cfg4 TP 8 rank GQA 8:1 B 64 ctx 2047)
#include <hip/hip_runtime.h>
__device__ long long* g_attn_trace = nullptr;
#define ATTN_STAMP_DECL long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; bool st3_ = false; st_[7] = __builtin_amdgcn_s_memrealtime(); \
    const long long xcc_ = __builtin_amdgcn_s_getreg((3 << 11) | 20) /* HW_REG_XCC_ID[3:0] */;
#define ATTN_STAMP(i)                                                        \
    do {                                                                     \
        if ((i) != 3 || !st3_) {                                             \
            st_[i] = __builtin_amdgcn_s_memtime();                           \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               \
            if ((i) == 3) st3_ = true;                                       \
        }                                                                    \
    } while (0)
#define ATTN_STAMP_FLUSH                                                                                                  \
    if (g_attn_trace && (threadIdx.x & 63) == 0) {                                                                        \
        long long* tp_ = g_attn_trace + ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 +    \
                                         (threadIdx.x >> 6)) * 16;                                                        \
        for (int i_ = 0; i_ < 8; ++i_) tp_[i_] = st_[i_];                                                                 \
        tp_[8] = xcc_;                                                                                                    \
    }
// ATTN_ROT=r: block (y, z) takes kv head (y + r) % gridDim.y — workgroups go to XCD (y + gridDim.y z) % 8, so r rotates which XCD
// reads which heads' 8 KiB slices of every page (are some XCD <-> address pairs closer than others?)
__device__ int g_attn_head_rot = 0;
#define ATTN_BLOCK_REMAP(by, bz) do { by = (by + g_attn_head_rot) % (int)gridDim.y; } while (0)
#include "attention.hip"

#include <algorithm>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint64_t rng_state = 0x9876543ull;
static inline uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }

static void run(const char* name, int B, int H, int Hkv, int D, int ctx, int dtype, int ns_override) {
    const int pages_per = (ctx + 31) / 32, total = B * pages_per, sets = 6;
    const size_t pool_bytes = (size_t)total * Hkv * 32 * D * 2;
    std::vector<void*> kp(sets), vp(sets);
    for (int i = 0; i < sets; ++i) {
        CK(hipMalloc(&kp[i], pool_bytes)); CK(hipMalloc(&vp[i], pool_bytes));
        CK(hipMemset(kp[i], 0x3c, pool_bytes)); CK(hipMemset(vp[i], 0x3c, pool_bytes));
    }
    std::vector<int32_t> bt(total), ctxl(B, ctx), cu(B + 1);
    if (getenv("ATTN_ORDER") && !strcmp(getenv("ATTN_ORDER"), "page-major")) {   // the product's pool since round 5: page p of every sequence side by side
        for (int b = 0; b < B; ++b) for (int p = 0; p < pages_per; ++p) bt[b * pages_per + p] = p * B + b;
    } else {                                                                        // rounds 1-4: a fixed pseudo-random order
        for (int i = 0; i < total; ++i) bt[i] = i;
        for (int i = total - 1; i > 0; --i) std::swap(bt[i], bt[rnd() % (i + 1)]);
    }
    for (int i = 0; i <= B; ++i) cu[i] = i;
    int32_t *dbt, *dctx, *dcu;
    CK(hipMalloc(&dbt, total * 4)); CK(hipMalloc(&dctx, B * 4)); CK(hipMalloc(&dcu, (B + 1) * 4));
    CK(hipMemcpy(dbt, bt.data(), total * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dctx, ctxl.data(), B * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice));
    void *q, *out;
    CK(hipMalloc(&q, (size_t)B * H * D * 2)); CK(hipMemset(q, 0x3c, (size_t)B * H * D * 2));
    CK(hipMalloc(&out, (size_t)B * H * D * 2));
    const int ns = ns_override > 0 ? ns_override : tgis_attn_num_splits(B, Hkv, H, 1, ctx);
    const int64_t wsb = tgis_attn_workspace_bytes(B, H, Hkv, D, ns);
    void* ws = nullptr;
    if (wsb) CK(hipMalloc(&ws, wsb));
    auto launch = [&](int i) {
        int rc = tgis_attn_paged(q, (int64_t)H * D, kp[i % sets], vp[i % sets], dbt, pages_per, dctx, dcu, out, 0, B, H, Hkv, D, 1, ctx,
                                 1.f / sqrtf((float)D), dtype, ns, ws, wsb, nullptr);
        if (rc != 0) { printf("tgis_attn_paged: %s\n", tgis_last_error()); exit(1); }
    };
    for (int i = 0; i < 4; ++i) launch(i);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 24; ++i) launch(i);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 1000.f / 24);
    }
    const double mb = (double)B * ctx * 2 * Hkv * D * 2 / 1e6;
    printf("%s: B %d H %d Hkv %d D %d ctx %d splits %d: %.1f us per launch (combine launch included where there is one), %.2f TB/s\n", name, B, H, Hkv, D,
           ctx, ns, best, mb / best);
    // the traced launch: grid and waves per block are the launcher's choice -> a generous buffer, zero = no wave there
    const size_t slots = (size_t)1 << 16;   // (block, wave) pairs
    long long* dtr; CK(hipMalloc(&dtr, slots * 16 * 8)); CK(hipMemset(dtr, 0, slots * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &dtr, sizeof(dtr)));
    launch(3);
    CK(hipDeviceSynchronize());
    long long* nul = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &nul, sizeof(nul)));
    std::vector<long long> tr(slots * 16);
    CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
    std::vector<size_t> live;
    for (size_t i = 0; i < slots; ++i) if (tr[i * 16 + 6] != 0) live.push_back(i);
    if (live.empty()) { printf("    (no stamps)\n"); return; }
    long long rt0 = tr[live[0] * 16 + 7];
    for (size_t i : live) rt0 = std::min(rt0, tr[i * 16 + 7]);
    {
        std::vector<long long> v;
        for (size_t i : live) v.push_back((tr[i * 16 + 7] - rt0) * 10);
        std::sort(v.begin(), v.end());
        printf("    %zu waves; wave entry after the first wave's entry (ns): median %lld  p90 %lld  max %lld\n", live.size(), v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    }
    const char* names[7] = {"entry", "lengths in", "first table entry in", "first page applied", "pages done", "O in LDS (barrier)", "stored"};
    printf("    ticks from the wave's own entry (min / median / max):\n");
    for (int k = 1; k < 7; ++k) {
        std::vector<long long> v;
        for (size_t i : live) v.push_back(tr[i * 16 + k] - tr[i * 16]);
        std::sort(v.begin(), v.end());
        printf("      %-22s %7lld %7lld %7lld\n", names[k], v[0], v[v.size() / 2], v.back());
    }
    {   // who finishes late?  'pages done' by XCD (block id % 8), by wave of the block, by dispatch order (block id quartile)
        auto med = [&](std::vector<long long>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0ll : v[v.size() / 2]; };
        size_t agree = 0;
        for (size_t i : live) agree += (size_t)tr[i * 16 + 8] == (i / 8) % 8;
        printf("    XCC_ID == linear block id %% 8 for %zu of %zu waves\n", agree, live.size());
        printf("    'pages done' medians by XCD (hardware id):");
        for (int x = 0; x < 8; ++x) {
            std::vector<long long> v;
            for (size_t i : live) if (tr[i * 16 + 8] == x) v.push_back(tr[i * 16 + 4] - tr[i * 16]);
            printf(" %lld", med(v));
        }
        printf("\n    by wave of the block:");
        for (int w = 0; w < 8; ++w) {
            std::vector<long long> v;
            for (size_t i : live) if (i % 8 == (size_t)w) v.push_back(tr[i * 16 + 4] - tr[i * 16]);
            if (!v.empty()) printf(" %lld", med(v));
        }
        const size_t nb = live.back() / 8 + 1;
        printf("\n    by block id quartile:");
        for (int qd = 0; qd < 4; ++qd) {
            std::vector<long long> v;
            for (size_t i : live) if ((i / 8) * 4 / nb == (size_t)qd) v.push_back(tr[i * 16 + 4] - tr[i * 16]);
            printf(" %lld", med(v));
        }
        printf("\n    absolute end ('stored' + entry offset, ns after the first entry; the tick rate from the slowest wave): ");
        std::vector<long long> v;
        for (size_t i : live) v.push_back((tr[i * 16 + 7] - rt0) * 10);
        printf("entry skew max %lld ns\n", *std::max_element(v.begin(), v.end()));
    }
    {   // the launch's own span on the 100 MHz clock: first entry -> (last wave's entry + its ticks to `stored`, at the tick rate below)
        std::vector<double> rate;
        printf("    pages per wave: ~%.1f;  per page between 'first page applied' and 'pages done' (median wave): ", (double)pages_per / ns / 1.0);
        std::vector<long long> v;
        for (size_t i : live) v.push_back(tr[i * 16 + 4] - tr[i * 16 + 3]);
        std::sort(v.begin(), v.end());
        printf("%lld ticks in total\n", v[v.size() / 2]);
    }
    CK(hipFree(dtr));
    for (int i = 0; i < sets; ++i) { CK(hipFree(kp[i])); CK(hipFree(vp[i])); }
    CK(hipFree(dbt)); CK(hipFree(dctx)); CK(hipFree(dcu)); CK(hipFree(q)); CK(hipFree(out));
    if (ws) CK(hipFree(ws));
}

int main(int argc, char** argv) {
    if (getenv("ATTN_ROT_SWEEP")) {   // cfg3 shape only, every rotation of the head <-> XCD assignment
        for (int r = 0; r < 8; ++r) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_head_rot), &r, sizeof(r)));
            char name[64]; snprintf(name, sizeof(name), "cfg3 MHA, heads rotated by %d", r);
            run(name, 32, 32, 32, 128, 1023, TGIS_F16, 0);
        }
        return 0;
    }
    run("cfg3 MHA", 32, 32, 32, 128, 1023, TGIS_F16, 0);
    run("cfg3 MHA (again)", 32, 32, 32, 128, 1023, TGIS_F16, 0);
    run("cfg3 MHA (ctx 1024)", 32, 32, 32, 128, 1024, TGIS_F16, 0);
    run("cfg5 MQA 48:1", 32, 48, 1, 128, 4095, TGIS_BF16, 0);
    run("cfg5 MQA 48:1 (8 splits)", 32, 48, 1, 128, 4095, TGIS_BF16, 8);
    run("cfg4 TP 8 rank GQA 8:1", 64, 8, 1, 128, 2047, TGIS_F16, 0);
    return 0;
}
