// Paged causal attention for prefill and decode on gfx950 (MFMA 16x16x32, wave64).
// Replaces flash_attn_2_cuda.varlen_fwd as called from utils/flash_attn.py:43-78 for both call shapes of
// custom_modeling/flash_llama_modeling.py:271-295 (prefill: causal over the fresh k/v; decode: q-len 1
// over all cached slots) — here both read the paged cache that tgis_rope_kv_write has just filled.
//
// One workgroup = 4 waves handles one (sequence, kv head, 16-column q tile, key split).  The 16 MFMA
// columns are (q token, q head of the GQA group) pairs, so one K/V stream serves the whole group.
// Groups wider than 16 heads (multi-query attention: 48 q heads on one kv head) take CH 16-head chunks per block:
// the wave keeps CH sets of Q fragments, softmax statistics and O accumulators and applies every K/V fragment it
// loads to all of them, so K/V still stream from HBM once (one block per chunk re-read them once per chunk — the
// chunks land on different XCDs, i.e. different L2s: 47 us per Starcoder-15B launch where the bytes need 11).
// Per 32-token page and wave:
//   S^T[tok][col] = K[tok][:] . Q[col][:]      A = K fragment (1 KiB contiguous loads), B = Q^T
//   online softmax per column: the column's statistics live in lane (l & 15) of every 16-lane row
//   O^T[d][col]  += V^T[d][tok] . P^T[tok][col] A = V^T fragment (1 KiB contiguous loads), B = P^T
// S^T's accumulator layout IS the B-operand layout of the second MFMA and O^T's column index is the
// same lane, so the only cross-lane traffic per page is the 2-step row-max exchange.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <map>
#include <mutex>
#include "common.h"
#include "attention_args.h"
#include "kv_layout.h"

// tools/floor/attn_unit.hip compiles this unit with per-wave s_memtime stamps (ATTN_STAMP*); the library does not.
#ifndef ATTN_STAMP
#define ATTN_STAMP_DECL
#define ATTN_STAMP(i)
#define ATTN_STAMP_FLUSH
#endif
#ifndef ATTN_BLOCK_REMAP  // tools/floor/attn_unit.hip: which (kv head, sequence) a block id takes, to probe XCD <-> address affinity
#define ATTN_BLOCK_REMAP(by, bz)
#endif

namespace {


// NW = waves per workgroup (the key range of a block is dealt page-by-page to its waves)
// PIPE: two pages of K/V loads in flight per wave (twice the fragment registers)
template <typename T, int D, int NW, int CH, bool PIPE>
__global__ __launch_bounds__(64 * NW) void attn_paged_kernel(AttnArgs a) {
    using V8 = typename VecT<T>::x8;
    constexpr int KS = D / 32;  // k-steps of the QK^T MFMA
    constexpr int NB = D / 16;  // 16-row blocks of O^T
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {   // every cache line of the argument block the decode path reads is requested at entry (one scalar round trip instead
        // of one per group of fields: tools/floor/wide.hip `pre`)
        const int64_t l0 = a.ld_q;
        const int l1 = a.NS;
        const void* l2 = a.counters;
        asm volatile("" ::"s"(l0), "s"(l1), "s"(l2));
    }
    ATTN_STAMP_DECL
    ATTN_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: the block-table reads become scalar loads
    const int col = lane & 15, c = lane >> 4;
    const int qt = blockIdx.x;
    int by = blockIdx.y, bz = blockIdx.z;
    ATTN_BLOCK_REMAP(by, bz);
    if (a.xcd_remap) {
        // Several blocks per (sequence, split, kv head) group — one per set of 16-head chunks — read the SAME K/V pages.
        // Workgroups go to the 8 XCDs round-robin by linear id, each XCD with its own L2: in grid order the chunk blocks of a
        // group land on different XCDs and every one of them pulls the pages from HBM.  Remapped, XCD x runs the blocks
        // x, x + 8, x + 16, ... and consecutive ones of them are the chunk blocks of one group: the pages cross the fabric
        // once and the other chunk blocks hit that XCD's L2 (MQA 48:1 as three one-chunk blocks, B = 32, ctx 4096:
        // 44 -> 33 us at 4 splits, profiles/r04_mqa_xcd.log).  gridDim.x == 1 and the number of groups is a multiple
        // of 8 — the launcher checks.
        const int L = blockIdx.y + gridDim.y * blockIdx.z;
        const int x = L & 7, i = L >> 3;
        const int grp = x + 8 * (i / a.HCB), cb = i % a.HCB;
        bz = grp / a.Hkv;
        by = (grp % a.Hkv) * a.HCB + cb;
    }
    const int hk = by / a.HCB, hc0 = (by % a.HCB) * CH;  // first 16-head chunk of this block
    const int b = bz / a.NS, split = bz % a.NS;

    // Unsplit launches (NS == 1: every block starts at page 0): this wave's first block-table entry does not depend on the
    // sequence's length, so it is requested together with the lengths instead of behind them — one dependent scalar round trip
    // less in front of the first K / V load (inside a decode step those lines are not in this XCD's L2: the GEMMs of the layer
    // have streamed ~100 MB through it since the last attention launch; profiles/r06_attn_cold.log).
    const int32_t* btrow = a.bt + (int64_t)b * a.max_pages;
    const int pg_spec = (a.NS == 1 && w < a.max_pages) ? btrow[w] : 0;
    const int q0 = a.cu_q[b], q_len = a.cu_q[b + 1] - q0;
    const int t0 = qt * a.TQ;
    if (t0 >= q_len) return;
    const int ctx = a.ctx_lens[b];
    const int tq = col >> a.Gp_shift, g = col & (a.Gp - 1);
    bool col_valid[CH];
    int kmax[CH];  // this column may attend to key positions < kmax
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        col_valid[ch] = (g < a.Gc) && ((hc0 + ch) * 16 + g < a.G) && (t0 + tq < q_len);
        kmax[ch] = col_valid[ch] ? (ctx - q_len + t0 + tq + 1) : 0;
    }
    const int kend = ctx - q_len + min(q_len, t0 + a.TQ);  // keys needed by any column of the tile
    const int pages = (kend + 31) >> 5;
    const int pps = (pages + a.NS - 1) / a.NS;
    const int pbeg = split * pps, pend = min(pages, pbeg + pps);
    ATTN_STAMP(1);  // (the sequence's lengths are in: the stamp macro waits for scalar loads)
    // The first fully visible page's table entry is asked for HERE — as soon as the lengths are in, in front of the q loads and
    // of the partly visible pages: behind them it was a scalar round trip with nothing of this wave in flight (round 5 timeline,
    // profiles/r05_attn_timeline.log: multi-chunk blocks had their first entry 4.8 k ticks after entry, 3.5 k after the lengths).
    const int pg_first = a.NS == 1 ? pg_spec : ((pbeg + w < pend) ? btrow[pbeg + w] : 0);

    // Q^T fragments (B operand): lane supplies Q[col][ks*32 + c*8 .. +8]
    V8 qf[CH][KS];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        const int head = hk * a.G + (hc0 + ch) * 16 + g;
        const T* qp = reinterpret_cast<const T*>(a.q) + (int64_t)(q0 + t0 + tq) * a.ld_q + (int64_t)head * D + c * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (col_valid[ch]) {
                qf[ch][ks] = ld16<V8>(qp + ks * 32);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[ch][ks][e] = (T)0.f;
            }
        }
    }
    // (Rotary embedding + cache write of the new token in this prologue was built in round 2, bit-exact, and measured slower
    // than the launch in front: cfg3 104 vs 93 us — experiments/README.md.)

    f32x4 o[CH][NB];
    float m[CH], lsum[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) o[ch][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        m[ch] = NEG_BIG;
        lsum[ch] = 0.f;
    }

    // one page of K (two 16-token halves x KS k-steps) and V^T (NB 16-row blocks): 16 KiB per wave at D = 128
    auto load_page = [&](const int pg, V8 (&kf)[2][KS], V8 (&vf)[NB]) {
        const T* kb = reinterpret_cast<const T*>(a.kpool) + ((int64_t)pg * a.Hkv + hk) * (32 * D) + lane * 8;
        const T* vb = reinterpret_cast<const T*>(a.vpool) + ((int64_t)pg * a.Hkv + hk) * (32 * D) + c * (D * 8) + col * 8;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[t][ks] = __builtin_nontemporal_load(
                reinterpret_cast<const V8*>(kb + t * (16 * D) + ks * 512));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) vf[nb] = __builtin_nontemporal_load(reinterpret_cast<const V8*>(vb + nb * 128));
    };
    // The page a sequence is still filling: only the part its nv written tokens occupy is requested (round 5: the launch
    // cost one whole page per started page — bench.py's steps got 0.1 ms slower the moment the batch crossed a page
    // boundary, and the counters saw 1.016 x the algorithmic bytes).  K: the second 16-token tile is requested only once the
    // sequence has reached it — until then its registers repeat the first tile (finite values, scores masked); a scalar base,
    // the lanes' offsets stay what they are.  V keeps tokens 4 g .. 4 g + 3 of both tiles in column group g and a lane reads one
    // group: the lanes of groups behind the last written token re-read group 0 (lines the wave requests anyway; finite values
    // under a P of exactly 0).  Addresses move, no lane is switched off.  (Redirecting the unwritten tokens INSIDE the tile that
    // is being filled as well costs 6 - 9 more VGPRs — two instead of three waves per SIMD, which the many-block shapes need;
    // tests/test_ops_gpu.py::test_attention_decode_every_fill_of_the_last_page poisons what must not be requested.)
    auto load_page_part = [&](const int pg, const int nv, V8 (&kf)[2][KS], V8 (&vf)[NB]) {
        const T* kp = reinterpret_cast<const T*>(a.kpool) + ((int64_t)pg * a.Hkv + hk) * (32 * D);
        const T* kp1 = kp + (nv > 16 ? 16 * D : 0);  // wave-uniform: a scalar base, the lanes' offsets stay what they are
        const T* vb = reinterpret_cast<const T*>(a.vpool) + ((int64_t)pg * a.Hkv + hk) * (32 * D) + col * 8 +
                      ((4 * c < nv) ? c * (D * 8) : 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[0][ks] = __builtin_nontemporal_load(reinterpret_cast<const V8*>(kp + lane * 8 + ks * 512));
            kf[1][ks] = __builtin_nontemporal_load(reinterpret_cast<const V8*>(kp1 + lane * 8 + ks * 512));
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) vf[nb] = __builtin_nontemporal_load(reinterpret_cast<const V8*>(vb + nb * 128));
    };
    // FULL: every key of the page is visible to every valid column (all but the last page of a decode step) — no
    // masks.  The softmax reference m[ch] is only moved when a tile maximum exceeds it by more than 2^RESCALE_LOG2
    // (then P <= 2^RESCALE_LOG2, harmless in fp32 / f16 / bf16): the rescale of O — whose accumulators live in AGPRs,
    // so every multiply costs a read and a write-back as well — and of the running sum then runs in a handful of
    // pages per sequence instead of every page, behind a wave-uniform branch.  (MQA, 3 chunks: 697 -> ~250 VALU
    // instructions per page; the loop is issue-bound, not HBM-bound: tools/attn_nw.py.)
    constexpr float RESCALE_LOG2 = 8.f;
    auto apply_page = [&](const int p, const V8 (&kf)[2][KS], const V8 (&vf)[NB], auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int kp0 = p * 32 + c * 4;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) s[t] = mfma16(kf[t][ks], qf[ch][ks], s[t]);
            }
            // scale, causal/length mask, tile max
            float tmax = NEG_BIG;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = s[t][r] * a.scale_log2;
                    if (!FULL) v = (kp0 + t * 16 + r < kmax[ch]) ? v : NEG_BIG;
                    s[t][r] = v;
                    tmax = fmaxf(tmax, v);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            if (__any(tmax > m[ch] + RESCALE_LOG2)) {
                const float m_new = fmaxf(m[ch], tmax);
                const float alpha = __builtin_amdgcn_exp2f(m[ch] - m_new);  // 0 while m is still NEG_BIG
                m[ch] = m_new;
                lsum[ch] *= alpha;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) o[ch][nb] *= alpha;
            }
            const float mref = m[ch];
            V8 pf;
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = __builtin_amdgcn_exp2f(s[t][r] - mref);
                    // masked entries contribute exactly 0 even while m is still NEG_BIG
                    if (!FULL) pv = (s[t][r] > 0.5f * NEG_BIG) ? pv : 0.f;
                    T pt = from_f32<T>(pv);
                    pf[t * 4 + r] = pt;
                    psum += to_f32(pt);  // normaliser from the rounded P, as flash-attention does
                }
            lsum[ch] += psum;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) o[ch][nb] = mfma16(vf[nb], pf, o[ch][nb]);
        }
        ATTN_STAMP(3);  // (kept once: the wave's first page applied)
    };
    // pages below kfull are visible in full to every column of the tile (the tile's first token sees kfull keys)
    const int kfull = ctx - q_len + t0 + 1;
    int p = pbeg + w;
    ATTN_STAMP(2);  // q requested, first table entry in
    // The partly visible pages of this wave (decode: the sequence's last page) go FIRST (round 4): the softmax is order-
    // independent, and their masked body is a second copy of the page code that each wave runs once — at the end of the
    // launch its instruction fetch and its un-overlapped load were part of every block's tail (ctx 1023 vs 1024: +2.6 us
    // per launch before, +1.7 after); at the start both hide under the first pages' latency.
    {
        const int nfullp = kfull >> 5;  // pages [0, nfullp) are fully visible
        int pp = p + ((max(nfullp - p, 0) + NW - 1) / NW) * NW;
        int pgp = (pp < pend) ? btrow[pp] : 0;
        for (; pp < pend; pp += NW) {
            const int pg_next = (pp + NW < pend) ? btrow[pp + NW] : 0;
            V8 kf[2][KS], vf[NB];
            load_page_part(pgp, min(32, ctx - pp * 32), kf, vf);
            apply_page(pp, kf, vf, std::false_type{});
            pgp = pg_next;
        }
    }
    if (PIPE) {
        // Pairs of fully visible pages with two pages of loads in flight: the next page's 16 KiB are requested before
        // the current page is applied.  The body is straight-line (every load it issues is needed, the trip count is
        // wave-uniform) so that hipcc keeps counted vmcnt waits; with a conditional prefetch it drains to vmcnt(0)
        // before every load and nothing overlaps (measured: no gain).  What is left — an odd full page — takes the plain
        // loop below (the partly visible pages ran first).
        const int nfull = (kfull >> 5) > p ? min(((kfull >> 5) - p + NW - 1) / NW, (pend - p + NW - 1) / NW) : 0;
        const int npairs = p < pend ? nfull >> 1 : 0;
        if (npairs > 0) {
            V8 kfa[2][KS], vfa[NB], kfb[2][KS], vfb[NB];
            load_page(pg_first, kfa, vfa);
            for (int i = 0; i + 1 < npairs; ++i) {
                load_page(btrow[p + NW], kfb, vfb);
                apply_page(p, kfa, vfa, std::true_type{});
                load_page(btrow[p + 2 * NW], kfa, vfa);
                apply_page(p + NW, kfb, vfb, std::true_type{});
                p += 2 * NW;
            }
            load_page(btrow[p + NW], kfb, vfb);
            apply_page(p, kfa, vfa, std::true_type{});
            apply_page(p + NW, kfb, vfb, std::true_type{});
            p += 2 * NW;
        }
    }
    // the fully visible pages — a loop of their own, not one loop with both bodies: the register allocator sizes a loop
    // for the union of what its branches hold (180 vs 122 VGPRs here, i.e. 2 instead of 3 waves per SIMD, which costs
    // the HBM-bound many-block shapes 5 %)
    int pg = PIPE ? ((p < pend) ? btrow[p] : 0) : pg_first;
    for (; p < pend && p * 32 + 32 <= kfull; p += NW) {
        const int pg_next = (p + NW < pend) ? btrow[p + NW] : 0;
        V8 kf[2][KS], vf[NB];
        load_page(pg, kf, vf);
        apply_page(p, kf, vf, std::true_type{});
        pg = pg_next;
    }

    ATTN_STAMP(4);  // pages done
    // ---- combine the 4 waves through LDS ----------------------------------------------------------
    float* so = reinterpret_cast<float*>(smem);          // [CH][NW][D][16]
    float* sml = so + CH * NW * D * 16;                  // [CH][NW][2][16]
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        float ls = lsum[ch];
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            // (row d = nb 16 + c 4 + r sits at nb 16 + r 4 + c: the four 16-lane groups of a store then hit four different
            // groups of 16 banks instead of the same one — round 5: the [d][16] order was a 4-way conflict on every access)
            for (int r = 0; r < 4; ++r) so[((ch * NW + w) * D + nb * 16 + r * 4 + c) * 16 + col] = o[ch][nb][r];
        if (c == 0) {
            sml[((ch * NW + w) * 2 + 0) * 16 + col] = m[ch];
            sml[((ch * NW + w) * 2 + 1) * 16 + col] = ls;
        }
    }
    __syncthreads();
    ATTN_STAMP(5);  // every wave's O and statistics in LDS
    const int grp = blockIdx.x + gridDim.x * (by + gridDim.y * b);  // the NS blocks that share their columns
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.ws_o, 0, 0x7FFFFFFF, 0x00020000);
    __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.ws_ml, 0, 0x7FFFFFFF, 0x00020000);
    // thread -> (chunk, column j, 8 consecutive d)
    for (int item = tid; item < CH * 16 * (D / 8); item += 64 * NW) {
        const int j = item & 15, dc = (item >> 4) % (D / 8), ch = item / (16 * (D / 8));
        const int hc = hc0 + ch;
        const int tqj = j >> a.Gp_shift, gj = j & (a.Gp - 1);
        if (!(gj < a.Gc && hc * 16 + gj < a.G && t0 + tqj < q_len)) continue;
        const float* soc = so + ch * NW * D * 16;
        const float* smc = sml + ch * NW * 2 * 16;
        float mw[NW], mstar = NEG_BIG;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            mw[k] = smc[(k * 2) * 16 + j];
            mstar = fmaxf(mstar, mw[k]);
        }
        float l = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            float f = exp2f(mw[k] - mstar);
            l += smc[(k * 2 + 1) * 16 + j] * f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = dc * 8 + e;
                acc[e] += soc[(k * D + ((d & ~15) | ((d & 3) << 2) | ((d >> 2) & 3))) * 16 + j] * f;
            }
        }
        const int64_t tokidx = q0 + t0 + tqj;
        const int headj = hk * a.G + hc * 16 + gj;
        if (a.NS == 1) {
            const float inv = l > 0.f ? 1.f / l : 0.f;
            V8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = from_f32<T>(acc[e] * inv);
            st16(reinterpret_cast<T*>(a.out) + (a.out_frag ? xf_off(tokidx, (int64_t)headj * D + dc * 8, (int64_t)a.H * D)
                                                           : (tokidx * a.H + headj) * D + dc * 8), ov);
        } else if (a.counters) {
            // fused combine: the record leaves as 16-byte write-through stores (no other block's record shares a
            // 128-byte line with it: 16 columns x NS x 16 bytes per (group, chunk))
            const int64_t rec = (((int64_t)grp * CH + ch) * 16 + j) * a.NS + split;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{acc[0], acc[1], acc[2], acc[3]}), ro,
                                                   (uint32_t)((rec * D + dc * 8) * 4), 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{acc[4], acc[5], acc[6], acc[7]}), ro,
                                                   (uint32_t)((rec * D + dc * 8 + 4) * 4), 0, 16);
            if (dc == 0)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{mstar, l, 0.f, 0.f}), rm,
                                                       (uint32_t)(rec * 16), 0, 16);
        } else {
            float* wo = a.ws_o + ((tokidx * a.H + headj) * a.NS + split) * D + dc * 8;
            *reinterpret_cast<f32x4*>(wo) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f32x4*>(wo + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
            if (dc == 0) {
                float* wm = a.ws_ml + ((tokidx * a.H + headj) * a.NS + split) * 2;
                wm[0] = mstar;
                wm[1] = l;
            }
        }
    }
    ATTN_STAMP(6);  // output or split record stored
    ATTN_STAMP_FLUSH
    if (a.NS == 1 || !a.counters) return;

    // ---- the last of the group's NS blocks to get here merges the NS records (what attn_combine_kernel does in a
    //      second launch otherwise).  Protocol of MI355X_MICROARCH.md: write-through (sc1) payload, drained, then an
    //      agent-scope counter; the reader uses sc1 loads on lines its XCD cannot hold yet. ------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* lastp = reinterpret_cast<int*>(smem);
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.counters + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old + 1u == (unsigned)a.NS;
        if (last) __hip_atomic_store(a.counters + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
        *lastp = last;
    }
    __syncthreads();
    if (!*lastp) return;
    // IB of this thread's items x MB splits are loaded before the first use (one round trip per batch instead of one
    // per split and item); batches are folded with the usual running-maximum rescale.
    constexpr int ITEMS = CH * 16 * (D / 8), THREADS = 64 * NW, ITER = (ITEMS + THREADS - 1) / THREADS;
    // <= 8 records (96 registers) in flight; the three-chunk (MQA) blocks, one per CU anyway, take 12: their four splits
    // arrive in one round trip instead of two
    constexpr int IB = ITER < 4 ? ITER : 4, MB = (CH == 3 && IB == 3 ? 12 : 8) / IB;
    for (int it0 = 0; it0 < ITER; it0 += IB) {
        bool ok[IB];
        int64_t rec0[IB];
        int dcs[IB];
        float mrun[IB], lrun[IB], acc[IB][8];
#pragma unroll
        for (int it = 0; it < IB; ++it) {
            const int item = tid + (it0 + it) * THREADS;
            const int j = item & 15, dc = (item >> 4) % (D / 8), ch = item / (16 * (D / 8));
            const int tqj = j >> a.Gp_shift, gj = j & (a.Gp - 1);
            ok[it] = item < ITEMS && gj < a.Gc && (hc0 + ch) * 16 + gj < a.G && t0 + tqj < q_len;
            rec0[it] = ok[it] ? (((int64_t)grp * CH + ch) * 16 + j) * a.NS : 0;  // (clamped: the loads stay in bounds)
            dcs[it] = dc;
            mrun[it] = NEG_BIG;
            lrun[it] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[it][e] = 0.f;
        }
        for (int s0 = 0; s0 < a.NS; s0 += MB) {
            f32x4 ml[IB][MB], lo[IB][MB], hi[IB][MB];
#pragma unroll
            for (int it = 0; it < IB; ++it)
#pragma unroll
                for (int k = 0; k < MB; ++k) {
                    const int64_t rec = rec0[it] + min(s0 + k, a.NS - 1);
                    ml[it][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, (uint32_t)(rec * 16), 0, 16));
                    lo[it][k] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(ro, (uint32_t)((rec * D + dcs[it] * 8) * 4), 0, 16));
                    hi[it][k] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(ro, (uint32_t)((rec * D + dcs[it] * 8 + 4) * 4), 0, 16));
                }
#pragma unroll
            for (int it = 0; it < IB; ++it) {
                float mb = mrun[it];
#pragma unroll
                for (int k = 0; k < MB; ++k)
                    if (s0 + k < a.NS) mb = fmaxf(mb, ml[it][k][0]);
                const float fr = exp2f(mrun[it] - mb);  // 0 for the first batch (mrun = NEG_BIG)
                mrun[it] = mb;
                lrun[it] *= fr;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[it][e] *= fr;
#pragma unroll
                for (int k = 0; k < MB; ++k) {
                    if (s0 + k < a.NS) {
                        const float f = exp2f(ml[it][k][0] - mb);
                        lrun[it] += ml[it][k][1] * f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[it][e] += lo[it][k][e] * f;
                            acc[it][e + 4] += hi[it][k][e] * f;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < IB; ++it) {
            if (!ok[it]) continue;
            const int item = tid + (it0 + it) * THREADS;
            const int j = item & 15, ch = item / (16 * (D / 8));
            const int tqj = j >> a.Gp_shift, gj = j & (a.Gp - 1);
            const float inv = lrun[it] > 0.f ? 1.f / lrun[it] : 0.f;
            V8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = from_f32<T>(acc[it][e] * inv);
            const int64_t tokidx = q0 + t0 + tqj;
            const int headj = hk * a.G + (hc0 + ch) * 16 + gj;
            st16(reinterpret_cast<T*>(a.out) + (a.out_frag ? xf_off(tokidx, (int64_t)headj * D + dcs[it] * 8, (int64_t)a.H * D)
                                                           : (tokidx * a.H + headj) * D + dcs[it] * 8), ov);
        }
    }
}

// out[tok][head][:] = sum_s O_s * 2^(m_s - m*) / sum_s l_s * 2^(m_s - m*);  one wave per (tok, head), four per workgroup.
// MAXS > 0 (NS <= MAXS): every record of the row — {m, l} and the O slice of each split — is requested before the first use,
// so the launch is ONE round trip to the records instead of two dependent passes over them (round 6: the MQA launch pair
// spends 4.5 of its 26 us here); MAXS == 0: any NS, the two-pass loop.  Same expressions in the same order either way.
template <typename T, int D, int MAXS>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ ws_o,
                                                           const float* __restrict__ ws_ml, T* __restrict__ out,
                                                           int NS, int H, int out_frag, int64_t rows) {
    const int64_t th = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // tok*H + head
    if (th >= rows) return;
    const int lane = threadIdx.x & 63;
    float mstar = NEG_BIG;
    float l = 0.f;
    float acc[D / 64] = {};
    if (MAXS > 0) {
        float ms[MAXS ? MAXS : 1], ls[MAXS ? MAXS : 1], os[MAXS ? MAXS : 1][D / 64];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const int64_t r = th * NS + min(s, NS - 1);
            ms[s] = ws_ml[r * 2];
            ls[s] = ws_ml[r * 2 + 1];
#pragma unroll
            for (int i = 0; i < D / 64; ++i) os[s][i] = ws_o[r * D + i * 64 + lane];
        }
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < NS) mstar = fmaxf(mstar, ms[s]);
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < NS) {
                const float f = exp2f(ms[s] - mstar);
                l += ls[s] * f;
#pragma unroll
                for (int i = 0; i < D / 64; ++i) acc[i] += os[s][i] * f;
            }
    } else {
        for (int s = 0; s < NS; ++s) mstar = fmaxf(mstar, ws_ml[(th * NS + s) * 2]);
        for (int s = 0; s < NS; ++s) {
            float f = exp2f(ws_ml[(th * NS + s) * 2] - mstar);
            l += ws_ml[(th * NS + s) * 2 + 1] * f;
#pragma unroll
            for (int i = 0; i < D / 64; ++i) acc[i] += ws_o[(th * NS + s) * D + i * 64 + lane] * f;
        }
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
    for (int i = 0; i < D / 64; ++i) {
        const int64_t o = out_frag ? xf_off(th / H, (th % H) * D + i * 64 + lane, (int64_t)H * D) : th * D + i * 64 + lane;
        out[o] = from_f32<T>(acc[i] * inv);
    }
}

static int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct AttnGeom {
    int G, Gc, Gp, TQ, HC;
};
static AttnGeom geom(int H, int Hkv) {
    AttnGeom g;
    g.G = H / Hkv;
    g.Gc = std::min(g.G, 16);
    g.HC = (g.G + 15) / 16;
    g.Gp = next_pow2(g.Gc);
    g.TQ = 16 / g.Gp;
    return g;
}

// Decode with fewer than 256 (sequence, kv head) groups: blocks of 8 waves that share one group's keys page by page
// (and merge through LDS) instead of more key splits merged across blocks.
static bool wide_decode_blocks(int64_t groups, int ch) { return ch == 1 && groups < 256; }

// 16-head chunks one decode block serves (register budget: 3 sets of O accumulators at D = 128)
static int chunks_per_block(int HC, int64_t max_q_len) {
    if (const char* e = getenv("TGIS_ATTN_CH")) {  // tuning hook (tools/attn_nw.py)
        const int v = atoi(e);
        if (v >= 1 && v <= 3) return (max_q_len == 1 && HC > 1) ? std::min(HC, v) : 1;
    }
    return (max_q_len == 1 && HC > 1) ? std::min(HC, 3) : 1;
}

template <typename T, int D, int NW, int CH, bool PIPE>
static void launch_attn_pipe(const AttnArgs& a, dim3 grid, hipStream_t st) {
    const size_t lds = (size_t)CH * (NW * D * 16 + NW * 2 * 16) * sizeof(float);
    static bool attr = false;  // e.g. CH = 3, D = 128, NW = 4: 98 KB of combine scratch
    if (!attr && lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)attn_paged_kernel<T, D, NW, CH, PIPE>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((attn_paged_kernel<T, D, NW, CH, PIPE>), grid, dim3(64 * NW), lds, st, a);
}

// Two pages in flight where a block is alone on its CU anyway (multi-chunk blocks: nothing else there hides the load
// latency; MQA 48:1, B=32, ctx 4096: 30.3 -> 25.8 us).  The 8-wave blocks of few-group shapes measured no different,
// and the many-block shapes keep the small-register variant.
template <typename T, int D, int NW, int CH>
static void launch_attn_one(const AttnArgs& a, dim3 grid, hipStream_t st) {
    static const int pipe_env = getenv("TGIS_ATTN_PIPE") ? atoi(getenv("TGIS_ATTN_PIPE")) : -1;  // A/B hook
    if constexpr (CH > 1 && NW == 4) {
        if (pipe_env != 0) return launch_attn_pipe<T, D, NW, CH, true>(a, grid, st);
    }
    if constexpr (CH == 1 && (NW == 2 || NW == 4)) {
        if (pipe_env == 2) return launch_attn_pipe<T, D, NW, CH, true>(a, grid, st);
    }
    launch_attn_pipe<T, D, NW, CH, false>(a, grid, st);
}

template <typename T, int D, int CH>
static void launch_attn_nw(const AttnArgs& a, dim3 grid, hipStream_t st, int nw) {
    if (nw == 1) return launch_attn_one<T, D, 1, CH>(a, grid, st);
    if (nw == 2) return launch_attn_one<T, D, 2, CH>(a, grid, st);
    if constexpr (CH == 1) {  // wide blocks: few (sequence, kv head) groups, the waves of one block share the keys
        if (nw == 8) return launch_attn_one<T, D, 8, CH>(a, grid, st);
        if (nw == 3) return launch_attn_one<T, D, 3, CH>(a, grid, st);  // three waves per SIMD at 1024 blocks (A/B hook)
    }
    launch_attn_one<T, D, 4, CH>(a, grid, st);
}

template <typename T, int D>
static int launch_attn(const AttnArgs& a, dim3 grid, int64_t total_q, hipStream_t st, int nw, int ch) {
    if (ch == 3)
        launch_attn_nw<T, D, 3>(a, grid, st, nw);
    else if (ch == 2)
        launch_attn_nw<T, D, 2>(a, grid, st, nw);
    else
        launch_attn_nw<T, D, 1>(a, grid, st, nw);
    TGIS_CHECK_LAUNCH();
    if (a.NS > 1 && !a.counters) {
        const int64_t rows = total_q * a.H;
        if (a.NS <= 8)
            hipLaunchKernelGGL((attn_combine_kernel<T, D, 8>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, st, a.ws_o,
                               a.ws_ml, (T*)a.out, a.NS, a.H, a.out_frag, rows);
        else
            hipLaunchKernelGGL((attn_combine_kernel<T, D, 0>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, st, a.ws_o,
                               a.ws_ml, (T*)a.out, a.NS, a.H, a.out_frag, rows);
        TGIS_CHECK_LAUNCH();
    }
    return TGIS_OK;
}

}  // namespace

extern "C" int tgis_attn_num_splits(int64_t B, int Hkv, int H, int64_t max_q_len, int64_t max_ctx) {
    if (B <= 0 || Hkv <= 0 || H <= 0 || H % Hkv != 0 || max_q_len != 1) return 1;  // splits: decode only
    AttnGeom g = geom(H, Hkv);
    int64_t q_tiles = cdiv64(max_q_len, g.TQ);
    const int ch = chunks_per_block(g.HC, max_q_len);
    int64_t base = B * Hkv * cdiv64(g.HC, ch) * q_tiles;
    int64_t pages = cdiv64(std::max<int64_t>(max_ctx, 1), 32);
    // single-chunk blocks, many groups: ~512 blocks fill 256 CUs twice; more, thinner blocks lose to the dispatch ramp
    // and the merge
    int64_t ns;
    if (wide_decode_blocks(base, ch)) {
        // few (sequence, kv head) groups: 8-wave blocks, one round of <= 256 of them, >= 2 pages per wave
        // (tools/attn_nw.py, us: B=16 GQA 8:1 D=64 ctx 512: 6.8 unsplit vs 8.5 at 4 splits of 4 waves;
        //  B=1 MHA D=128 ctx 2048: 14.1 at 8 splits vs 17.7 at 16; B=4 GQA 4:1 ctx 4096: 19.0 at 8 vs 24.0 at 16)
        ns = std::min<int64_t>(cdiv64(256, base), pages / 16);
        // Round 6 (profiles/r06_attn_fewgroups.log, first column = the rule above): from 32 pages on, up to 8 splits of >= 8
        // pages pay as long as every block still gets a CU of its own — B 1 MHA ctx 1024 11.6 -> 9.5 us (2 -> 8 splits), B 4
        // GQA 8:1 12.3 -> 10.4, B 8 GQA 8:1 D 64 9.4 -> 8.3, B 2 MHA 12.9 -> 11.8; 96 groups keep 2 (3 would be 288 blocks:
        // 16.4 vs 14.3), 16 pages keep 1 (8.7 vs 10.2), long contexts keep the rule above (>= 16 pages per block).
        if (pages >= 32) ns = std::max<int64_t>(ns, std::min<int64_t>(std::min<int64_t>(256 / base, pages / 8), 8));
        // From 128 groups on the merge costs more than the second half of the CUs gives while the context is short (a 7B rank at
        // TP = 8, B 32 x 4 heads, ctx 1024: 16.0 us unsplit, 16.5 at 2 splits; 64 groups, ctx 2048: 24.0 / 19.2 / 17.7 at
        // 1 / 2 / 4 — profiles/r05_attn_tp8.log).  Round 6 swept it over the context (profiles/r06_attn_base128.log): exactly
        // 128 groups take 2 splits from ctx 2048 on (26.9 vs 29.1 us, 46.0 vs 51.7 at 4096, 166.6 vs 194.8 at 16384: the page
        // walk grows, the merge does not); 129 - 255 groups never do (2 splits would be a second round of blocks: 192 groups
        // 41.2 vs 34.9 at 2048, 272.8 vs 243.5 at 16384).
        if (base >= 128) ns = (base * 2 <= 256 && pages >= 64) ? 2 : 1;
    } else if (ch > 1) {
        // multi-chunk blocks hold three sets of accumulators: one block per CU, one wave per SIMD, so the block's time is
        // its waves' page count (~1 us per page: issue-bound) plus ~12 us.  Their splits are merged by the combine launch,
        // not in the launch (a last-arriving block reads NS x 24 KB through ONE CU: 2.7 us at 4 splits, 5.4 at 8).
        // 48 q heads on 1 kv head, B=32 ctx 4096, us (kernel + combine): 26.7 at 4 splits, 25.1 at 6, 25.3 at 8
        // (in-launch merge: 28.1 / 28.6 / 30.2)
        // round 6: with the combine launch down to one round trip (24.1 vs 25.3 us at 6 splits) 8 splits = 256 blocks edge out
        // 6 (23.9 vs 24.2, profiles/r06_mqa_variants.log)
        ns = std::min<int64_t>(cdiv64(256, base), pages / 16);
    } else {
        ns = cdiv64(512, base);
        ns = std::min<int64_t>(ns, cdiv64(pages, 4));  // at least one page per wave
    }
    ns = std::max<int64_t>(1, std::min<int64_t>(ns, 64));
    return (int)ns;
}

// records of the fused combine: [group = (sequence, kv head, chunk block)][chunk][16 columns][split]
static int64_t fused_records(int64_t B, int H, int Hkv, int num_splits) {
    AttnGeom g = geom(H, Hkv);
    const int ch = chunks_per_block(g.HC, 1);
    return B * Hkv * cdiv64(g.HC, ch) * ch * 16 * num_splits;
}

extern "C" int64_t tgis_attn_workspace_bytes(int64_t total_q_tokens, int H, int Hkv, int D, int num_splits) {
    if (num_splits <= 1 || Hkv <= 0 || H % Hkv != 0) return 0;
    // the larger of the two layouts: per (token, head, split) {O[D], m, l} for the two-launch combine, or the
    // line-padded records of the fused one (16 columns per group and chunk, {m, l} in 16 bytes)
    const int64_t two_pass = total_q_tokens * H * num_splits * ((int64_t)D + 2) * 4;
    const int64_t fused = fused_records(total_q_tokens, H, Hkv, num_splits) * ((int64_t)D + 4) * 4;
    return std::max(two_pass, fused);
}

namespace {
constexpr int64_t ATTN_COUNTERS = 1 << 20;
constexpr int ATTN_COUNTER_SETS = 4;

// Arrival counters of the fused combine, one per group; zero between launches (the last arriver resets its own).
// Owned by the library: a pool of ATTN_COUNTER_SETS arrays per device, handed out per STREAM — two split-key launches that
// overlap on one device are necessarily on different streams, so they never share an array (launches of one stream are
// ordered).  The pool is allocated by the first call that is not inside a stream capture (the eager warm-up step); handing
// an array of the pool to a new stream needs no allocation, so a capture stream gets one too.  A fifth concurrent stream,
// or a capture before any eager call, falls back to the separate combine launch.
struct CounterPool {
    unsigned* sets[ATTN_COUNTER_SETS] = {};
    hipStream_t owner[ATTN_COUNTER_SETS] = {};
    uint64_t last_use[ATTN_COUNTER_SETS] = {};  // launch stamp: the least recently used slot is handed to a new stream
    bool in_graph[ATTN_COUNTER_SETS] = {};      // a captured graph holds this slot's pointer: it never changes hands
    int pending[ATTN_COUNTER_SETS] = {};        // handed out, launch not enqueued yet: an idle owner stream proves nothing
    uint64_t stamp = 0;
    int used = 0;
    bool ready = false;
    bool warned = false;
};
std::mutex g_attn_counters_mu;
std::map<int, CounterPool> g_attn_counters;

// `slot` receives the index to give back with attn_counters_enqueued() once the launch that uses the array is in its
// stream (between this call and the launch a second host thread could see the new owner's stream idle and take
// the same array).
unsigned* attn_counters(hipStream_t st, int* dev_out, int* slot) {
    *slot = -1;
    static const bool off = getenv("TGIS_ATTN_FUSED_COMBINE") && atoi(getenv("TGIS_ATTN_FUSED_COMBINE")) == 0;
    if (off) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    *dev_out = dev;
    std::lock_guard<std::mutex> lock(g_attn_counters_mu);
    CounterPool& pool = g_attn_counters[dev];
    if (!pool.ready) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;
        }
        unsigned* p = nullptr;
        const size_t bytes = (size_t)ATTN_COUNTER_SETS * ATTN_COUNTERS * sizeof(unsigned);
        if (hipMalloc((void**)&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        (void)hipDeviceSynchronize();
        for (int i = 0; i < ATTN_COUNTER_SETS; ++i) pool.sets[i] = p + (size_t)i * ATTN_COUNTERS;
        pool.ready = true;
    }
    ++pool.stamp;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool query_failed = hipStreamIsCapturing(st, &cs) != hipSuccess;
    if (query_failed) (void)hipGetLastError();  // not this launch's error: do not leave it for TGIS_CHECK_LAUNCH
    const bool capturing = query_failed || cs != hipStreamCaptureStatusNone;
    for (int i = 0; i < pool.used; ++i)
        if (pool.owner[i] == st) {
            pool.last_use[i] = pool.stamp;
            pool.in_graph[i] |= capturing;
            ++pool.pending[i];
            *slot = i;
            return pool.sets[i];
        }
    if (pool.used < ATTN_COUNTER_SETS) {
        pool.owner[pool.used] = st;
        pool.last_use[pool.used] = pool.stamp;
        pool.in_graph[pool.used] = capturing;
        ++pool.pending[pool.used];
        *slot = pool.used;
        return pool.sets[pool.used++];
    }
    // All slots are owned.  A slot whose stream has no launch in flight can change hands (its counters are zero between
    // launches): take the least recently used one if its stream is idle (hipStreamQuery on a destroyed or capturing stream
    // fails — then the slot stays put and this launch uses the separate combine kernel, which is always correct).
    // (a slot whose pointer sits in a captured graph stays where it is: the graph replays on streams this pool never sees)
    int lru = -1;
    for (int i = 0; i < ATTN_COUNTER_SETS; ++i)
        if (!pool.in_graph[i] && pool.pending[i] == 0 && (lru < 0 || pool.last_use[i] < pool.last_use[lru])) lru = i;
    hipStreamCaptureStatus ocs = hipStreamCaptureStatusNone;
    // (a query on a stream that is being captured would invalidate its capture: ask that first)
    const bool owner_capturing = lru < 0 || hipStreamIsCapturing(pool.owner[lru], &ocs) != hipSuccess ||
                                 ocs != hipStreamCaptureStatusNone;
    if (!capturing && !owner_capturing && hipStreamQuery(pool.owner[lru]) == hipSuccess) {
        pool.owner[lru] = st;
        pool.last_use[lru] = pool.stamp;
        ++pool.pending[lru];
        *slot = lru;
        return pool.sets[lru];
    }
    (void)hipGetLastError();
    if (!pool.warned) {
        pool.warned = true;
        fprintf(stderr, "tgis_attn_paged: more than %d streams use split-key decode attention on device %d; the extra ones "
                        "merge their splits with a second launch\n", ATTN_COUNTER_SETS, dev);
    }
    return nullptr;
}
void attn_counters_enqueued(int dev, int slot) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_attn_counters_mu);
    --g_attn_counters[dev].pending[slot];
}
struct CounterLease {  // gives the slot back when the launch is in its stream (or the call fails before it)
    int dev = 0, slot = -1;
    ~CounterLease() { attn_counters_enqueued(dev, slot); }
};
}  // namespace

static int attn_paged_impl(const void* q, int64_t ld_q, const void* k_pool, const void* v_pool,
                           const int32_t* block_tables, int64_t max_pages, const int32_t* ctx_lens,
                           const int32_t* cu_seqlens_q, void* out, int64_t ld_out, int64_t B, int H, int Hkv, int D,
                           int64_t max_q_len, int64_t max_ctx, float scale, int dtype, int num_splits,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    TGIS_CHECK_ARG(q && k_pool && v_pool && block_tables && ctx_lens && cu_seqlens_q && out,
                   "tgis_attn_paged: null tensor");
    TGIS_CHECK_ARG(H > 0 && Hkv > 0 && H % Hkv == 0, "tgis_attn_paged: H (%d) must be a multiple of Hkv (%d)", H, Hkv);
    TGIS_CHECK_ARG(D == 64 || D == 128, "tgis_attn_paged: head_dim %d not supported (64, 128)", D);
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "tgis_attn_paged: bad dtype");
    TGIS_CHECK_ARG(ld_q % 8 == 0 && ((uintptr_t)q % 16) == 0, "tgis_attn_paged: q must be 16-byte aligned");
    TGIS_CHECK_ARG(max_q_len > 0 && max_pages > 0 && num_splits >= 1, "tgis_attn_paged: bad launch bounds");
    TGIS_CHECK_ARG(num_splits == 1 || max_q_len == 1, "tgis_attn_paged: key splits are for decode (max_q_len == 1)");
    TGIS_CHECK_ARG(ld_out == 0 || ld_out == (int64_t)H * D ||
                       (ld_out == TGIS_LD_FRAGMENTS && max_q_len == 1 && B <= 64 && ((int64_t)H * D) % 64 == 0),
                   "tgis_attn_paged: out is [tokens, H * D] contiguous (ld_out = 0 or H * D), or — decode, <= 64 sequences — in "
                   "fragment order (ld_out = TGIS_LD_FRAGMENTS)");
    (void)max_ctx;
    if (B == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    AttnGeom g = geom(H, Hkv);
    AttnArgs a;
    a.q = q;
    a.ld_q = ld_q;
    a.kpool = k_pool;
    a.vpool = v_pool;
    a.bt = block_tables;
    a.max_pages = max_pages;
    a.ctx_lens = ctx_lens;
    a.cu_q = cu_seqlens_q;
    a.out = out;
    a.out_frag = ld_out == TGIS_LD_FRAGMENTS;
    a.H = H;
    a.Hkv = Hkv;
    a.G = g.G;
    a.Gc = g.Gc;
    a.Gp = g.Gp;
    a.Gp_shift = 0;
    while ((1 << a.Gp_shift) < g.Gp) ++a.Gp_shift;
    a.TQ = g.TQ;
    a.HC = g.HC;
    const int ch = chunks_per_block(g.HC, max_q_len);
    a.HCB = (g.HC + ch - 1) / ch;
    a.NS = num_splits;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.ws_o = nullptr;
    a.ws_ml = nullptr;
    a.counters = nullptr;
    int64_t total_q = 0;
    CounterLease lease;
    if (num_splits > 1) {
        // total q tokens is only needed to size the split workspace; callers pass B*max_q_len rows
        total_q = B * max_q_len;
        int64_t need = tgis_attn_workspace_bytes(total_q, H, Hkv, D, num_splits);
        TGIS_CHECK_ARG(workspace && workspace_bytes >= need, "tgis_attn_paged: workspace too small (%ld < %ld)",
                       (long)workspace_bytes, (long)need);
        a.ws_o = (float*)workspace;
        a.ws_ml = a.ws_o + total_q * H * num_splits * D;
        // one launch: the last block of each (sequence, kv head) group merges the splits (no combine launch)
        const int64_t recs = fused_records(B, H, Hkv, num_splits);
        const int64_t groups = B * Hkv * a.HCB;
        if (groups <= ATTN_COUNTERS && recs * D * 4 < (1ll << 31) && ch == 1) {
            a.counters = attn_counters(st, &lease.dev, &lease.slot);
            if (a.counters) a.ws_ml = a.ws_o + recs * D;
        }
    }
    // long q (prefill): blocks of 128 columns that stage each K/V page once in LDS (attention_prefill.hip)
    if (num_splits == 1 && max_q_len * g.Gp > 64 && !getenv("TGIS_ATTN_NO_PREFILL_KERNEL")) {
        TgisTimedScope timed(TGIS_OP_ATTN, st);
        return tgis_launch_attn_prefill(a, B, Hkv, D, max_q_len, dtype, st);
    }
    int64_t q_tiles = cdiv64(max_q_len, g.TQ);
    TGIS_CHECK_ARG(q_tiles <= 2147483647LL && (int64_t)Hkv * a.HCB <= 65535 && B * num_splits <= 65535,
                   "tgis_attn_paged: grid too large");
    dim3 grid((unsigned)q_tiles, (unsigned)(Hkv * a.HCB), (unsigned)(B * num_splits));
    {
        static const bool xcd_off = getenv("TGIS_ATTN_XCD") && atoi(getenv("TGIS_ATTN_XCD")) == 0;  // A/B hook
        a.xcd_remap = (!xcd_off && max_q_len == 1 && q_tiles == 1 && a.HCB > 1 && (B * num_splits * Hkv) % 8 == 0) ? 1 : 0;
    }
    // waves per block: with >= 1024 (sequence, kv head) blocks the chip is full either way and 2-wave blocks halve
    // the page-count imbalance between a block's waves (33 pages over 4 waves = 9/8/8/8; over 2 = 17/16)
    const int64_t nblocks = (int64_t)grid.x * grid.y * grid.z;
    // (past ~2k blocks the dispatch rate, 7 ns per workgroup, costs more than the imbalance)
    int nw = (max_q_len == 1 && nblocks >= 1024 && nblocks < 2048) ? 2 : 4;
    if (max_q_len == 1 && wide_decode_blocks(nblocks / num_splits, ch)) nw = 8;
    if (const char* e = getenv("TGIS_ATTN_NW")) {
        const int v = atoi(e);
        nw = (v == 1 || v == 2 || v == 3 || v == 8) ? v : 4;
        if (ch > 1 && (nw > 4 || nw == 3)) nw = 4;
    }
    TgisTimedScope timed(TGIS_OP_ATTN, st);
    if (dtype == TGIS_F16) {
        if (D == 128) return launch_attn<f16, 128>(a, grid, total_q, st, nw, ch);
        return launch_attn<f16, 64>(a, grid, total_q, st, nw, ch);
    } else {
        if (D == 128) return launch_attn<bf16, 128>(a, grid, total_q, st, nw, ch);
        return launch_attn<bf16, 64>(a, grid, total_q, st, nw, ch);
    }
}

extern "C" int tgis_attn_paged(const void* q, int64_t ld_q, const void* k_pool, const void* v_pool,
                               const int32_t* block_tables, int64_t max_pages, const int32_t* ctx_lens,
                               const int32_t* cu_seqlens_q, void* out, int64_t ld_out, int64_t B, int H, int Hkv, int D,
                               int64_t max_q_len, int64_t max_ctx, float scale, int dtype, int num_splits,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    return attn_paged_impl(q, ld_q, k_pool, v_pool, block_tables, max_pages, ctx_lens, cu_seqlens_q, out, ld_out, B, H, Hkv, D,
                           max_q_len, max_ctx, scale, dtype, num_splits, workspace, workspace_bytes, stream);
}

