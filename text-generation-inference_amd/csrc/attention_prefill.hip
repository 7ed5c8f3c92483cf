// Prefill attention over the paged cache (gfx950, MFMA 16x16x32): the many-q-rows counterpart of attn_paged_kernel.
// Same mathematics, page layouts and column convention (a column = one (q token, q head of the GQA group) pair);
// what changes is who reads K/V.  In attn_paged_kernel every wave loads the fragments of a page for its own 16
// columns, which is right for decode (one pass over the cache) and wasteful for prefill, where the same page is needed
// by every q tile of the sequence: measured 100-180 TFLOP/s, bound by L1 fragment traffic.  Here a block of 4 waves
// owns 128 columns (32 per wave, two 16-column groups), stages each 32-token page ONCE in LDS (double-buffered, one
// barrier per page) and every wave takes its K / V^T fragments from there: one global read per page and block, and each
// fragment read from LDS feeds two MFMAs.
//
// LDS images of a page (16 KiB): K as in global memory ([tile][D/8][16 tokens][8]: a fragment is one contiguous KiB,
// lane l reads piece l); V^T regrouped to [D/16][4 chunks][16 rows][8 token columns] so that lane (row l&15, chunk
// l>>4) again reads piece l of a contiguous KiB (the global [4 chunks][D][8] order would spread a fragment over four runs).
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "attention_args.h"

namespace {

template <typename T, int D>
__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(AttnArgs a) {
    using V8 = typename VecT<T>::x8;
    constexpr int KS = D / 32;  // k-steps of the QK^T MFMA
    constexpr int NB = D / 16;  // 16-row blocks of O^T
    constexpr int CG = 2;       // 16-column groups per wave
    constexpr int PAGE_ELEMS = 32 * D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* lds = reinterpret_cast<T*>(smem);  // [2 buffers][K page | V page]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 15, c = lane >> 4;
    const int qt = gridDim.x - 1 - blockIdx.x;  // long (late) tiles first: the causal triangle's tail is short ones
    const int hk = blockIdx.y / a.HC, hc = blockIdx.y % a.HC;
    const int b = blockIdx.z;

    const int q0 = a.cu_q[b], q_len = a.cu_q[b + 1] - q0;
    const int TQB = a.TQ * 4 * CG;  // q tokens of the block's tile (128 columns)
    const int t0 = qt * TQB;
    if (t0 >= q_len) return;
    const int ctx = a.ctx_lens[b];
    const int g = col & (a.Gp - 1);
    const int head = hk * a.G + hc * 16 + g;
    int tq[CG], kmax[CG];
    bool col_valid[CG];
    int wave_kmax = 0, wave_kmin = 0x7fffffff;
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
        tq[cg] = ((w * CG + cg) * 16 + col) >> a.Gp_shift;
        col_valid[cg] = (g < a.Gc) && (hc * 16 + g < a.G) && (t0 + tq[cg] < q_len);
        kmax[cg] = col_valid[cg] ? (ctx - q_len + t0 + tq[cg] + 1) : 0;  // this column attends to positions < kmax
        wave_kmax = max(wave_kmax, kmax[cg]);
        if (col_valid[cg]) wave_kmin = min(wave_kmin, kmax[cg]);  // invalid columns (q = 0, never stored) need no mask
    }
    // wave-uniform bound: pages at or past it are fully masked for this wave and skipped
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        wave_kmax = max(wave_kmax, __shfl_xor(wave_kmax, o, 64));
        wave_kmin = min(wave_kmin, __shfl_xor(wave_kmin, o, 64));
    }
    const int kend = ctx - q_len + min(q_len, t0 + TQB);  // keys needed by any column of the block
    const int pages = (kend + 31) >> 5;

    // Q^T fragments (B operand): lane supplies Q[col][ks*32 + c*8 .. +8]
    V8 qf[CG][KS];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
        const T* qp = reinterpret_cast<const T*>(a.q) + (int64_t)(q0 + t0 + tq[cg]) * a.ld_q + (int64_t)head * D + c * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (col_valid[cg]) {
                qf[cg][ks] = ld16<V8>(qp + ks * 32);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[cg][ks][e] = (T)0.f;
            }
        }
    }

    f32x4 o[CG][NB];
    float m[CG], lsum[CG];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) o[cg][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        m[cg] = NEG_BIG;
        lsum[cg] = 0.f;
    }

    // ---- page staging: 2*PAGE_ELEMS/8 16-byte pieces per page, NP per thread ---------------------------------
    constexpr int NP = (2 * PAGE_ELEMS / 8) / 256;  // 4 (D = 64) or 8 (D = 128)
    const int32_t* btrow = a.bt + (int64_t)b * a.max_pages;
    V8 stage[NP];
    auto page_load = [&](int p) {
        const int pg = btrow[p];
        const T* kp = reinterpret_cast<const T*>(a.kpool) + ((int64_t)pg * a.Hkv + hk) * PAGE_ELEMS;
        const T* vp = reinterpret_cast<const T*>(a.vpool) + ((int64_t)pg * a.Hkv + hk) * PAGE_ELEMS;
#pragma unroll
        for (int i = 0; i < NP / 2; ++i) {
            stage[i] = ld16<V8>(kp + (i * 256 + tid) * 8);
            stage[NP / 2 + i] = ld16<V8>(vp + (i * 256 + tid) * 8);
        }
    };
    auto page_store = [&](int buf) {
        T* kl = lds + buf * 2 * PAGE_ELEMS;
        T* vl = kl + PAGE_ELEMS;
#pragma unroll
        for (int i = 0; i < NP / 2; ++i) {
            const int piece = i * 256 + tid;
            st16(kl + piece * 8, stage[i]);
            // global V^T piece = (chunk cc, row d) ([4][D][8]); LDS piece = (d/16)*64 + cc*16 + d%16
            const int d = piece % D, cc = piece / D;
            st16(vl + (((d >> 4) * 64 + cc * 16 + (d & 15)) * 8), stage[NP / 2 + i]);
        }
    };

    page_load(0);
    page_store(0);
    __syncthreads();

    for (int p = 0; p < pages; ++p) {
        const bool more = p + 1 < pages;
        if (more) page_load(p + 1);
        if (p * 32 < wave_kmax) {
            const T* kl = lds + (p & 1) * 2 * PAGE_ELEMS + lane * 8;
            const T* vl = kl + PAGE_ELEMS;
            const int kp0 = p * 32 + c * 4;
            f32x4 s[CG][2];
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int t = 0; t < 2; ++t) s[cg][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const V8 kf = ld16<V8>(kl + t * (16 * D) + ks * 512);
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg) s[cg][t] = mfma16(kf, qf[cg][ks], s[cg][t]);
                }
            V8 pf[CG];
            float alpha[CG];
            // Interior pages (every key visible to every column of the wave) take the mask-free form; a wave meets at
            // most ceil(32 columns / 32) + 1 diagonal pages per tile.
            auto softmax_page = [&](auto masked_tag) {
                constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    float tmax = NEG_BIG;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = s[cg][t][r] * a.scale_log2;
                            if (MASKED) v = (kp0 + t * 16 + r < kmax[cg]) ? v : NEG_BIG;
                            s[cg][t][r] = v;
                            tmax = fmaxf(tmax, v);
                        }
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                    const float m_new = fmaxf(m[cg], tmax);
                    alpha[cg] = __builtin_amdgcn_exp2f(m[cg] - m_new);
                    m[cg] = m_new;
                    float psum = 0.f;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float pv = __builtin_amdgcn_exp2f(s[cg][t][r] - m_new);
                            // masked entries contribute exactly 0 even while m is still NEG_BIG
                            if (MASKED) pv = (s[cg][t][r] > 0.5f * NEG_BIG) ? pv : 0.f;
                            const T pt = from_f32<T>(pv);
                            pf[cg][t * 4 + r] = pt;
                            psum += to_f32(pt);  // normaliser from the rounded P, as flash-attention does
                        }
                    lsum[cg] = lsum[cg] * alpha[cg] + psum;
                }
            };
            if ((p + 1) * 32 <= wave_kmin)
                softmax_page(std::false_type{});
            else
                softmax_page(std::true_type{});
            // the running maximum moves rarely once a few pages are in: rescale O only when some column's did
            const bool rescale = __builtin_amdgcn_ballot_w64(alpha[0] != 1.f || alpha[CG - 1] != 1.f) != 0;
            if (rescale) {
#pragma unroll
                for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) o[cg][nb] *= alpha[cg];
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const V8 vf = ld16<V8>(vl + nb * 512);
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) o[cg][nb] = mfma16(vf, pf[cg], o[cg][nb]);
            }
        }
        if (more) page_store((p + 1) & 1);
        __syncthreads();  // next page staged; everyone is done reading the buffer that the iteration after overwrites
    }

    // ---- epilogue: O^T[d][col] -> out[token][head][d], through a private LDS slice per wave -----------------
    float* so = reinterpret_cast<float*>(smem) + w * (D * 16);  // [D][16] floats (the page buffers are dead)
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
        float ls = lsum[cg];
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);  // every lane of a column now holds the column's normaliser
        const float inv = ls > 0.f ? 1.f / ls : 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) so[(nb * 16 + c * 4 + r) * 16 + col] = o[cg][nb][r] * inv;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slice: no barrier, only the wave's own LDS order
        for (int item = lane; item < 16 * (D / 8); item += 64) {
            const int j = item & 15, dc = item >> 4;
            const int tqj = ((w * CG + cg) * 16 + j) >> a.Gp_shift, gj = j & (a.Gp - 1);
            if (!(gj < a.Gc && hc * 16 + gj < a.G && t0 + tqj < q_len)) continue;
            V8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = from_f32<T>(so[(dc * 8 + e) * 16 + j]);
            st16(reinterpret_cast<T*>(a.out) + ((int64_t)(q0 + t0 + tqj) * a.H + hk * a.G + hc * 16 + gj) * D + dc * 8, ov);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

}  // namespace

int tgis_launch_attn_prefill(const AttnArgs& a, int64_t B, int Hkv, int D, int64_t max_q_len, int dtype, hipStream_t st) {
    const int TQB = a.TQ * 8;
    const int64_t q_tiles = cdiv64(max_q_len, TQB);
    TGIS_CHECK_ARG(q_tiles <= 2147483647LL && (int64_t)Hkv * a.HC <= 65535 && B <= 65535, "tgis_attn_paged: grid too large");
    dim3 grid((unsigned)q_tiles, (unsigned)(Hkv * a.HC), (unsigned)B);
    const size_t lds = (size_t)2 * 2 * 32 * D * 2;  // two buffers of (K page + V page); >= 4 wave slices of D*16 floats
#define TGIS_PREFILL_LAUNCH(T, DD) hipLaunchKernelGGL((attn_prefill_kernel<T, DD>), grid, dim3(256), lds, st, a)
    if (dtype == TGIS_F16) {
        if (D == 128) TGIS_PREFILL_LAUNCH(f16, 128); else TGIS_PREFILL_LAUNCH(f16, 64);
    } else {
        if (D == 128) TGIS_PREFILL_LAUNCH(bf16, 128); else TGIS_PREFILL_LAUNCH(bf16, 64);
    }
#undef TGIS_PREFILL_LAUNCH
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}
