// GPTQ int4 linear for gfx950: one-time repack ("prepare"), fused dequant + MFMA GEMM for decode-sized
// M, and a full dequant kernel for the large-M (library GEMM) path.
//
// Replaces exllamav2_kernels.make_q_matrix / gemm_half_q_half as called from
// utils/gptq/exllamav2.py:14-62,124-144.  Normative arithmetic (utils/gptq/quant_linear.py:130-138,
// 184-194):  W[k,n] = (q[k,n] - (z[g(k),n] + 1)) * s[g(k),n];  y = x @ W, fp32 accumulate, f16 out.
//
// Prepared image, NT = ceil(N/32) column tiles, KS = ceil(K/64) k-steps, G groups:
//   A: wq [NT][KS][64 lanes][4] int32 — lane l word i = the 8 nibbles of rows
//        k = (ks*8 + (l>>5)*4 + i)*8 + {0,2,4,6,1,3,5,7} of column n = nt*32 + (l&31): one KiB per wave
//        load, and (after dequant8) exactly the B-operand fragment of v_mfma_f32_32x32x16_f16
//   B: sz [NT][G][32] u32 = { scale f16 , (1024 + z + 1) f16 }
// Rows are pre-permuted by the act-order permutation when g_idx is not trivial.
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include <mutex>
#include "common.h"
#include <type_traits>
#include "gptq_gemm_body.h"
#include "gptq_wide_body.h"

namespace {

using gptq::PrepLayout;
using gptq::prep_layout;
using gptq::GemmArgs;
using gptq::GemmPlan;
using gptq::plan_gemm;
using gptq::slab_bytes;
using gptq::KC;
using gptq::RS;

// Column held by lane c (0..31) of tile nt.  flags bit 0 (gate/up interleave, for the fused SiLU*mul epilogue):
// lanes 0-15 hold gate columns 16 nt + c, lanes 16-31 the matching up columns N/2 + 16 nt + (c - 16).
// flags bit 1 (rope image of a fused qkv projection, for the rotary + cache-write epilogue; head size D in bits 8..19,
// number of rotated heads H + Hkv in bits 20..31): a tile of a rotated head holds dims [16 t, 16 t + 16) in lanes 0-15 and
// their rotation partners D/2 + [16 t, 16 t + 16) in lanes 16-31.
__device__ __forceinline__ int64_t col_src(int64_t nt, int c, int64_t N, int flags) {
    if (flags & 1) return (c < 16) ? nt * 16 + c : (N >> 1) + nt * 16 + (c - 16);
    if (flags & 2) {
        const int D = (flags >> 8) & 0xFFF, nrot = (flags >> 20) & 0xFFF, per = D >> 5;
        const int64_t head = nt / per, t = nt - head * per;
        if (head < nrot) return head * D + ((c < 16) ? 16 * t + c : (D >> 1) + 16 * t + (c - 16));
    }
    return nt * 32 + c;
}

__device__ __forceinline__ int nib_src(int j) {  // stored nibble j holds row offset {0,2,4,6,1,3,5,7}[j]
    return (j < 4) ? 2 * j : 2 * (j - 4) + 1;
}

__global__ void gptq_prepare_w_kernel(const int32_t* __restrict__ qweight, const int32_t* __restrict__ perm,
                                      int32_t* __restrict__ wq, int64_t K, int64_t N, int64_t NT, int64_t KS,
                                      int flags) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = NT * KS * 256;
    if (idx >= total) return;
    int i = idx & 3;
    int l = (idx >> 2) & 63;
    int64_t ks = (idx >> 8) % KS;
    int64_t nt = (idx >> 8) / KS;
    int64_t n = col_src(nt, l & 31, N, flags);
    int64_t p = ks * 8 + (l >> 5) * 4 + i;  // k-pack row (8 k each)
    uint32_t v = 0;
    if (n < N && p * 8 < K) {
        if (perm == nullptr) {
            uint32_t w = (uint32_t)qweight[p * N + n];
#pragma unroll
            for (int j = 0; j < 8; ++j) v |= ((w >> (4 * nib_src(j))) & 15u) << (4 * j);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int64_t ksrc = perm[p * 8 + nib_src(j)];
                uint32_t w = (uint32_t)qweight[(ksrc >> 3) * N + n];
                v |= ((w >> (4 * (ksrc & 7))) & 15u) << (4 * j);
            }
        }
    }
    wq[idx] = (int32_t)v;
}

__global__ void gptq_prepare_sz_kernel(const int32_t* __restrict__ qzeros, const f16* __restrict__ scales,
                                       uint32_t* __restrict__ sz, int64_t N, int64_t NT, int64_t G, int flags) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NT * G * 32) return;
    int c = idx & 31;
    int64_t g = (idx >> 5) % G;
    int64_t nt = (idx >> 5) / G;
    int64_t n = col_src(nt, c, N, flags);
    f16x2 v = {(f16)0.f, (f16)1025.f};
    if (n < N) {
        uint32_t w = (uint32_t)qzeros[g * (N / 8) + (n >> 3)];
        v[0] = scales[g * N + n];
        v[1] = (f16)(float)(1024u + ((w >> (4 * (n & 7))) & 15u) + 1u);
    }
    sz[idx] = __builtin_bit_cast(uint32_t, v);
}

template <int TN, int WK, int ACT, bool GROUP64, bool PERM, int MR>
__global__ __launch_bounds__(64 * TN * WK) void gptq_gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gptq::gptq_gemm_unit<TN, WK, ACT, GROUP64, PERM, MR>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// Decode batches of up to 32 rows whose activation is in fragment order (gptq_wide_body.h).  8 waves; CT = 4 holds 2 waves
// per SIMD (one block per CU), CT = 2 / 3 leave room for more.
template <int CT, int ACT, bool OUTF, int MR>
__global__ __launch_bounds__(64 * gptq::WIDE_WK) void gptq_wide_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gptq::gptq_wide_unit<CT, ACT, OUTF, MR>(a, smem);
}

// ---- "tall" kernel: 64 < M (decode batches beyond 32 rows, add-on prefills of up to a few thousand tokens) --------
// The streaming kernel above re-streams and re-dequantises the weights once per 32 (64) rows; the library path
// (dequantise the whole matrix to a scratch copy, then hipBLASLt) only pays off for very tall M.  In between, one block
// keeps BM = 32 BMR rows of x in LDS per 128-k chunk and each of its four waves owns one 32-column tile: a wave loads
// 1 KiB of the same prepared image per k64-step, dequantises it ONCE into four MFMA B fragments and applies them to
// all BMR row blocks (BMR x 4 MFMAs 32x32x16 per 52 dequantisation instructions: MFMA-bound from BMR = 2).  No scratch
// copy of W, no second pass over the weights, same epilogues (bias, SiLU * up on the interleaved gate/up image,
// split-K slabs in the 32-row units the consumers expect).
constexpr int TKC = 128;       // k per x chunk (2 k64-steps)
constexpr int TRS = TKC + 8;   // LDS row stride in halves (+16 B: conflict-free ds_read_b128 of A fragments)

// TW = 32-column tiles per wave.  With one tile every MFMA needs its own 1 KiB A fragment from LDS: eight resident waves x
// one ds_read_b128 per 32-cycle MFMA is the whole LDS bandwidth of the CU.  With two tiles an A fragment feeds two MFMAs
// (256-column blocks, half the LDS traffic per flop); the kernel is held to 256 registers so that two blocks still share
// a CU.
template <int BMR, int ACT, int TW>
__global__ __launch_bounds__(256, 2) void gptq_gemm_tall_kernel(GemmArgs a) {
    constexpr int BM = 32 * BMR;
    constexpr int NJ = BM * 16 / 256;  // 16-byte x pieces per thread and chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16* xs = reinterpret_cast<f16*>(smem);  // [2][BM][TRS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * BM;
    const int mrows = min(BM, a.M - m0);
    const int split = blockIdx.z;
    const int nt_raw0 = (blockIdx.x * 4 + w) * TW;
    const int ksteps = a.K >> 6;                     // K % 64 == 0 (host)
    const int per = a.KR >> 6;                       // k64-steps per split (multiple of 2)
    const int ks0 = split * per, ks1 = min(ksteps, ks0 + per);
    const int nchunks = (ks1 - ks0 + 1) >> 1;        // block-uniform

    const char* wtile[TW];
    const char* sztile[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int nt = min(nt_raw0 + t, a.NT - 1);
        wtile[t] = reinterpret_cast<const char*>(a.prep) + (int64_t)nt * a.KS * 1024;
        sztile[t] = reinterpret_cast<const char*>(a.prep + a.offB) + (int64_t)nt * a.G * 128;
    }
    const uint32_t woff = lane * 16, szoff = (lane & 31) * 4;
    auto sz_at = [&](int t, int ks) -> uint32_t {
        const int g = min(ks >> a.spg_shift, a.G - 1);
        const char* p = sztile[t] + (int64_t)g * 128;
        PIN_SGPR(p);
        const uint32_t v = *(const GLOBAL_AS uint32_t*)(p + szoff);
        return ks < ks1 ? v : 0u;  // steps past the split's range add zeros
    };
    auto w_at = [&](int t, int ks) -> u32x4 {
        const char* p = wtile[t] + (int64_t)min(ks, ksteps - 1) * 1024;
        PIN_SGPR(p);
        return __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(p + woff));
    };

    // x staging: piece p = tid + 256 j -> row p / 16, 8-element column (p % 16) * 8 of the chunk
    const f16* xbase = a.x + (int64_t)m0 * a.ldx;
    f16x8 xg[NJ];
    uint32_t rowoff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rowoff[j] = (uint32_t)(min((tid + 256 * j) >> 4, mrows - 1) * (int)a.ldx * 2);
    const int scol = (tid & 15) * 8;
    auto stage_load = [&](int chunk) {
        const int kc = min((ks0 << 6) + chunk * TKC + scol, a.K - 8);
        const char* xb = reinterpret_cast<const char*>(xbase);
        PIN_SGPR(xb);
#pragma unroll
        for (int j = 0; j < NJ; ++j) xg[j] = *(const GLOBAL_AS f16x8*)(xb + rowoff[j] + (uint32_t)kc * 2);
    };
    auto stage_store = [&](int buf) {
        f16* dst = xs + buf * (BM * TRS) + (tid >> 4) * TRS + scol;
#pragma unroll
        for (int j = 0; j < NJ; ++j) st16(dst + j * 16 * TRS, xg[j]);
    };

    uint32_t EXr = 0x64006400u, M0r = 0x000F000Fu, M1r = 0x00F000F0u;
    asm volatile("" : "+v"(EXr));
    asm volatile("" : "+s"(M0r), "+s"(M1r));
    f32x16 acc[TW][BMR];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int rb = 0; rb < BMR; ++rb) acc[t][rb] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int xoff = (lane & 31) * TRS + (lane >> 5) * 32;

    stage_load(0);
    u32x4 wq[TW][4];
    uint32_t szr[TW][4];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) szr[t][s] = sz_at(t, ks0 + s);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < TW; ++t) wq[t][s] = w_at(t, ks0 + s);
    stage_store(0);
    __syncthreads();

    // one chunk = 2 k64-steps in ring slots SB, SB + 1; the slots are refilled in place two chunks ahead
    auto chunk_body = [&](const int chunk, auto sb_tag) {
        constexpr int SB = decltype(sb_tag)::value;
        const bool more = chunk + 1 < nchunks;  // block-uniform
        if (more) stage_load(chunk + 1);
        uint32_t szn[TW][2];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) szn[t][s2] = sz_at(t, ks0 + chunk * 2 + s2 + 4);
        const f16* xbuf = xs + (chunk & 1) * (BM * TRS) + xoff;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ks = ks0 + chunk * 2 + s2;
            f16x8 b[TW][4];
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const u32x4 cur = wq[t][SB + s2];
                const f16x2 szh = __builtin_bit_cast(f16x2, szr[t][SB + s2]);
                const f16 zc1 = szh[1];
                const f16 zd1 = (f16)960.f - zc1;
                const f16x2 zc = {zc1, zc1}, zd = {zd1, zd1}, sc = {szh[0], szh[0]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    b[t][i] = gptq::dequant8(cur[i], zc, zd, sc, EXr, M0r, M1r);
                }
                wq[t][SB + s2] = w_at(t, ks + 4);
            }
            const f16* xk = xbuf + s2 * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int rb = 0; rb < BMR; ++rb) {
                    const f16x8 av = ld16<f16x8>(xk + rb * (32 * TRS) + i * 8);
#pragma unroll
                    for (int t = 0; t < TW; ++t) acc[t][rb] = mfma32(av, b[t][i], acc[t][rb]);
                }
        }
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) szr[t][SB + s2] = szn[t][s2];
        if (more) stage_store((chunk + 1) & 1);
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    int chunk = 0;
    for (; chunk + 1 < nchunks; chunk += 2) {
        chunk_body(chunk, I0{});
        chunk_body(chunk + 1, I2{});
    }
    if (chunk < nchunks) chunk_body(chunk, I0{});

    // ---- epilogue: lane holds rows m = 32 rb + (r&3) + 8 (r>>2) + 4 (lane>>5) of column n = nt * 32 + (lane & 31) ----
    const int c = lane & 31;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
    const int nt = nt_raw0 + t;
    if (nt >= a.NT) break;
    if (ACT == 2) {
        const int half = a.N >> 1;
        const int j = nt * 16 + (c & 15);
        const int nsrc = (c < 16) ? j : half + j;
        const float bv = a.bias ? (float)a.bias[nsrc] : 0.f;
#pragma unroll
        for (int rb = 0; rb < BMR; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float mine = (float)(f16)(acc[t][rb][r] + bv);
                const float other = __shfl_xor(mine, 16, 64);
                const int m = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (c < 16 && m < mrows) {
                    const float sl = mine / (1.f + __expf(-mine));
                    a.out[(int64_t)(m0 + m) * a.ldo + j] = (f16)((float)(f16)sl * other);
                }
            }
        continue;
    }
    const int n = nt * 32 + c;
    if (a.S == 1 && !a.partial) {
        const float bv = a.bias ? (float)a.bias[n] : 0.f;
#pragma unroll
        for (int rb = 0; rb < BMR; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < mrows) a.out[(int64_t)(m0 + m) * a.ldo + n] = (f16)(acc[t][rb][r] + bv);
            }
    } else {
        // slabs in 32-row units [unit][S][32][NP], unit = row / 32 (what norm / rope / the reduce kernel index)
        const int64_t np = (int64_t)a.NT * 32;
#pragma unroll
        for (int rb = 0; rb < BMR; ++rb) {
            const int unit = blockIdx.y * BMR + rb;
            if (unit * 32 >= a.M) continue;
            float* sl = a.slabs + ((int64_t)(unit * a.S + split) * 32) * np + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                sl[(int64_t)m * np] = acc[t][rb][r];
            }
        }
    }
    }
}

// Sum the S split-K slabs in fixed order (deterministic) and emit f16 (+bias): thread = (row, 4 columns).
__global__ __launch_bounds__(256) void splitk_reduce_f16_kernel(const float* __restrict__ slabs,
                                                                const f16* __restrict__ bias, f16* __restrict__ out,
                                                                int64_t ldo, int M, int N, int NP, int S) {
    const int np4 = NP >> 2;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int mslab = blockIdx.y;
    if (idx >= (int64_t)32 * np4) return;
    const int m = idx / np4, c4 = (idx - (int64_t)m * np4) * 4;
    if (mslab * 32 + m >= M) return;
    f32x4 v = {0, 0, 0, 0};
    const float* base = slabs + ((int64_t)mslab * S * 32 + m) * NP + c4;
    for (int s2 = 0; s2 < S; ++s2) v += *reinterpret_cast<const f32x4*>(base + (int64_t)s2 * 32 * NP);
    f16* o = out + (int64_t)(mslab * 32 + m) * ldo + c4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (c4 + e < N) {
            float f = v[e];
            if (bias) f += (float)bias[c4 + e];
            o[e] = (f16)f;
        }
    }
}

// The same sum for a gate | up image (tile nt holds gate columns 16 nt .. +15 and the matching up columns), followed by
// out[m][j] = f16(f16(silu(f16 gate)) * f16 up): the act = 2 epilogue of the GEMM, applied here when the projection ran
// split over k (64-row passes of a narrow shard: an unsplit block would take in its whole 64 x K activation).
__global__ __launch_bounds__(256) void splitk_reduce_silu_kernel(const float* __restrict__ slabs,
                                                                 const f16* __restrict__ bias, f16* __restrict__ out,
                                                                 int64_t ldo, int M, int N, int NP, int S) {
    const int half = N >> 1, nt4 = (NP >> 5) * 4;  // per row: NT tiles x 4 groups of 4 gate columns
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int mslab = blockIdx.y;
    if (idx >= (int64_t)32 * nt4) return;
    const int m = idx / nt4, q = idx - (int64_t)m * nt4, nt = q >> 2, c4 = (q & 3) * 4;
    if (mslab * 32 + m >= M) return;
    f32x4 g = {0, 0, 0, 0}, u = {0, 0, 0, 0};
    const float* base = slabs + ((int64_t)mslab * S * 32 + m) * NP + nt * 32 + c4;
    for (int s2 = 0; s2 < S; ++s2) {
        g += *reinterpret_cast<const f32x4*>(base + (int64_t)s2 * 32 * NP);
        u += *reinterpret_cast<const f32x4*>(base + (int64_t)s2 * 32 * NP + 16);
    }
    f16* o = out + (int64_t)(mslab * 32 + m) * ldo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = nt * 16 + c4 + e;
        if (j >= half) continue;
        const float mine = (float)(f16)(g[e] + (bias ? (float)bias[j] : 0.f));
        const float other = (float)(f16)(u[e] + (bias ? (float)bias[half + j] : 0.f));
        const float sl = mine / (1.f + __expf(-mine));
        o[j] = (f16)((float)(f16)sl * other);
    }
}

}  // namespace

// shared with gptq_lean.hip (declared in gptq_gemm_body.h)
int gptq::reduce_slabs(const float* slabs, const f16* bias, f16* out, int64_t ldo, int M, int N, int NP, int S,
                       hipStream_t st) {
    dim3 rgrid((unsigned)cdiv64((int64_t)32 * (NP / 4), 256), (unsigned)cdiv64(M, 32));
    hipLaunchKernelGGL(splitk_reduce_f16_kernel, rgrid, dim3(256), 0, st, slabs, bias, out, ldo, M, N, NP, S);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

namespace {

__global__ void gptq_dequant_kernel(const uint8_t* __restrict__ prep, int64_t offB, f16* __restrict__ wout,
                                    int K, int N, int G, int gs, int NT, int KS, int flags) {
    // one thread per prepared int32 (8 k of one column)
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)NT * KS * 256) return;
    int i = idx & 3;
    int l = (idx >> 2) & 63;
    int64_t ks = (idx >> 8) % KS;
    int64_t nt = (idx >> 8) / KS;
    int n = (int)col_src(nt, l & 31, N, flags);
    int k0 = (ks * 8 + (l >> 5) * 4 + i) * 8;
    if (n >= N || k0 >= K) return;
    uint32_t q = reinterpret_cast<const uint32_t*>(prep)[idx];
    int g = min(k0 / gs, G - 1);
    f16x2 szh = __builtin_bit_cast(f16x2, reinterpret_cast<const uint32_t*>(prep + offB)[(nt * G + g) * 32 + (l & 31)]);
    float s = (float)szh[0];
    float z = (float)szh[1] - 1024.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = ((float)((q >> (4 * j)) & 15u) - z) * s;
        wout[(int64_t)(k0 + nib_src(j)) * N + n] = (f16)v;
    }
}

}  // namespace

extern "C" int64_t tgis_gptq_prepared_bytes(int64_t K, int64_t N, int64_t groups) {
    if (K <= 0 || N <= 0 || groups <= 0) return 0;
    return prep_layout(K, N, groups).total;
}

extern "C" int tgis_gptq_prepare(const int32_t* qweight, const int32_t* qzeros, const void* scales,
                                 const int32_t* g_idx_host, int32_t* perm_out, int64_t K, int64_t N,
                                 int64_t groups, int flags, void* prepared, void* stream) {
    TGIS_CHECK_ARG(qweight && qzeros && scales && prepared, "tgis_gptq_prepare: null tensor");
    TGIS_CHECK_ARG(!(flags & 1) || (N % 32 == 0), "tgis_gptq_prepare: gate/up interleave needs N/2 %% 16 == 0");
    if (flags & 2) {
        const int D = (flags >> 8) & 0xFFF, nrot = (flags >> 20) & 0xFFF;
        TGIS_CHECK_ARG(!(flags & 1) && D >= 32 && D % 32 == 0 && nrot >= 1 && (int64_t)nrot * D <= N && N % D == 0,
                       "tgis_gptq_prepare: rope image needs head size %% 32 == 0 and rotated heads within N (D=%d, heads=%d)",
                       D, nrot);
    }
    TGIS_CHECK_ARG(K > 0 && N > 0 && K % 32 == 0 && N % 32 == 0,
                   "tgis_gptq_prepare: K (%ld) and N (%ld) must be positive multiples of 32", (long)K, (long)N);
    TGIS_CHECK_ARG(groups > 0 && K % groups == 0, "tgis_gptq_prepare: K %% groups != 0");
    const int64_t gs = K / groups;
    TGIS_CHECK_ARG(gs % 8 == 0, "tgis_gptq_prepare: group size %ld not a multiple of 8", (long)gs);
    hipStream_t st = (hipStream_t)stream;
    const int32_t* perm_dev = nullptr;
    if (g_idx_host) {
        bool trivial = true;
        for (int64_t k = 0; k < K; ++k)
            if (g_idx_host[k] != (int32_t)(k / gs)) { trivial = false; break; }
        if (!trivial) {
            TGIS_CHECK_ARG(perm_out, "tgis_gptq_prepare: act-order g_idx needs perm_out");
            std::vector<int32_t> perm(K);
            std::iota(perm.begin(), perm.end(), 0);
            std::stable_sort(perm.begin(), perm.end(),
                             [&](int32_t a, int32_t b) { return g_idx_host[a] < g_idx_host[b]; });
            // every group must own exactly gs rows (true for GPTQ act-order checkpoints)
            for (int64_t k = 0; k < K; ++k)
                TGIS_CHECK_ARG(g_idx_host[perm[k]] == (int32_t)(k / gs),
                               "tgis_gptq_prepare: g_idx groups are not of uniform size");
            TGIS_CHECK_HIP(hipStreamSynchronize(st));
            TGIS_CHECK_HIP(hipMemcpy(perm_out, perm.data(), K * sizeof(int32_t), hipMemcpyHostToDevice));
            perm_dev = perm_out;
        }
    }
    PrepLayout p = prep_layout(K, N, groups);
    uint8_t* base = (uint8_t*)prepared;
    int64_t totalA = p.NT * p.KS * 256;
    hipLaunchKernelGGL(gptq_prepare_w_kernel, dim3((unsigned)cdiv64(totalA, 256)), dim3(256), 0, st, qweight,
                       perm_dev, (int32_t*)base, K, N, p.NT, p.KS, flags);
    TGIS_CHECK_LAUNCH();
    int64_t totalB = p.NT * groups * 32;
    hipLaunchKernelGGL(gptq_prepare_sz_kernel, dim3((unsigned)cdiv64(totalB, 256)), dim3(256), 0, st, qzeros,
                       (const f16*)scales, (uint32_t*)(base + p.offB), N, p.NT, groups, flags);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

// ---- tall kernel: when, and how it is cut ---------------------------------------------------------------------------
// Rows from which the tall kernel replaces the 64-row streaming passes, and up to which it replaces dequantise + library
// GEMM.  tools/tall_sweep.py on MI355X, cfg3 shapes (profiles/r02_tall_sweep.log; us, tall / passes / library incl. its
// dequantisation): gate_up M=256 75 / 82 / 172, M=512 132 / 164 / 206, M=1024 235 / - / 304, M=2048 437 / - / 508, M=3072
// 640 / - / 659, M=4096 814 / - / 871; qkv M=512 81 / 82 / 100, M=1024 133 / - / 138, M=2048 227 / - / 230, M=3072 339 / -
// / 331, M=4096 423 / - / 381; the four GEMMs of a layer together: M=2048 954 / - / 1027, M=3072 1444 / - / 1373, M=4096
// 1752 / - / 1713, M=8192 3441 / - / 3056.  Up to 256 rows the passes (with their deferred split-K sums) stay; from 257 to
// 3072 rows the tall kernel runs at 0.65-0.9 PFLOP/s without a scratch copy of W; above, hipBLASLt on the dequantised
// copy (1.1-1.2 PFLOP/s) wins by a few per cent and more.  Two 32-column tiles per wave (an A fragment from LDS feeds two
// MFMAs; 1.0 PFLOP/s at M >= 4096) once 384 blocks of 256 columns remain; one tile per wave below.
static int64_t tall_min_m() {
    static const int64_t v = getenv("TGIS_TALL_MIN_M") ? atoll(getenv("TGIS_TALL_MIN_M")) : 257;
    return v;
}
static int64_t tall_max_m() {
    static const int64_t v = getenv("TGIS_TALL_MAX_M") ? atoll(getenv("TGIS_TALL_MAX_M")) : 3072;
    return v;
}
static bool tall_ok(int64_t M, int64_t K, int64_t groups, const int32_t* perm, int act) {
    if (M < tall_min_m() || perm != nullptr || act == 1 || K % 64 != 0) return false;
    const int64_t gs = K / groups, spg = gs / 64;
    return groups == 1 || (gs % 64 == 0 && (spg & (spg - 1)) == 0);
}
struct TallPlan {
    int BMR, S, KR, TW;
};
static TallPlan plan_tall(int64_t M, int64_t K, int64_t N, int act) {
    TallPlan t;
    t.BMR = M > 64 ? 4 : 2;
    // two tiles per wave (256-column blocks) once that still leaves every CU a few blocks
    t.TW = (t.BMR == 4 && cdiv64(cdiv64(N, 32), 8) * cdiv64(M, 128) >= 384) ? 2 : 1;
    if (const char* e = getenv("TGIS_TALL_TW")) t.TW = (atoi(e) == 2 && t.BMR == 4) ? 2 : 1;
    const int64_t blocks = cdiv64(cdiv64(N, 32), 4 * t.TW) * cdiv64(M, 32 * t.BMR);
    const int64_t kchunks = cdiv64(K, TKC);
    int64_t S = 1;
    if (act != 2 && blocks < 384) S = std::min<int64_t>(std::min<int64_t>(kchunks, 16), cdiv64(512, blocks));
    if (const char* e = getenv("TGIS_TALL_SPLITS")) S = std::max<int64_t>(1, std::min<int64_t>(kchunks, atoll(e)));
    if (act == 2) S = 1;
    int64_t per = cdiv64(kchunks, S);
    while (S > 1 && (S - 1) * per >= kchunks) --S;
    t.S = (int)S;
    t.KR = (int)(per * TKC);
    return t;
}
static int64_t tall_slab_bytes(int64_t M, int64_t N, int S) { return (int64_t)cdiv64(M, 32) * S * 32 * cdiv64(N, 32) * 32 * 4; }

static int launch_tall(const void* x, int64_t ldx, const void* prepared, const void* bias, void* out, int64_t ldo,
                       int64_t M, int64_t K, int64_t N, int64_t groups, int act, float* slabs, int partial,
                       const TallPlan& tp, hipStream_t st) {
    PrepLayout p = prep_layout(K, N, groups);
    const int64_t gs = K / groups, spg = gs / 64;
    GemmArgs a;
    a.x = (const f16*)x;
    a.ldx = ldx;
    a.prep = (const uint8_t*)prepared;
    a.offB = p.offB;
    a.bias = partial ? nullptr : (const f16*)bias;
    a.perm = nullptr;
    a.out = (f16*)out;
    a.ldo = ldo;
    a.M = (int)M;
    a.K = (int)K;
    a.N = (int)N;
    a.G = (int)groups;
    a.gs = (int)gs;
    a.KR = tp.KR;
    a.S = tp.S;
    a.NT = (int)p.NT;
    a.KS = (int)p.KS;
    a.slabs = slabs;
    a.partial = partial;
    a.spg_shift = 30;
    a.positions = a.slots = nullptr;
    a.cosb = a.sinb = nullptr;
    a.kpool = a.vpool = nullptr;
    a.rH = a.rHkv = a.rD = 0;
    if (groups > 1)
        for (a.spg_shift = 0; (1 << a.spg_shift) < spg; ++a.spg_shift) {}
    const int BM = 32 * tp.BMR;
    TGIS_CHECK_ARG(cdiv64(M, BM) <= 65535, "tgis_gptq_gemm: M too large for one launch");
    dim3 grid((unsigned)cdiv64(p.NT, 4 * tp.TW), (unsigned)cdiv64(M, BM), (unsigned)tp.S);
    const size_t lds = (size_t)2 * BM * TRS * sizeof(f16);
#define TGIS_TALL(B, A, W)                                                                                     \
    do {                                                                                                      \
        static bool attr = false;                                                                             \
        if (!attr) {                                                                                          \
            TGIS_CHECK_HIP(hipFuncSetAttribute((const void*)gptq_gemm_tall_kernel<B, A, W>,                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * TRS * 2)); \
            attr = true;                                                                                      \
        }                                                                                                     \
        hipLaunchKernelGGL((gptq_gemm_tall_kernel<B, A, W>), grid, dim3(256), lds, st, a);                    \
    } while (0)
    if (tp.BMR == 4 && tp.TW == 2) {
        if (act == 2) TGIS_TALL(4, 2, 2); else TGIS_TALL(4, 0, 2);
    } else if (tp.BMR == 4) {
        if (act == 2) TGIS_TALL(4, 2, 1); else TGIS_TALL(4, 0, 1);
    } else {
        if (act == 2) TGIS_TALL(2, 2, 1); else TGIS_TALL(2, 0, 1);
    }
#undef TGIS_TALL
    TGIS_CHECK_LAUNCH();
    if (!partial && tp.S > 1) {
        const int NP = (int)p.NT * 32;
        dim3 rgrid((unsigned)cdiv64((int64_t)32 * (NP / 4), 256), (unsigned)cdiv64(M, 32));
        hipLaunchKernelGGL(splitk_reduce_f16_kernel, rgrid, dim3(256), 0, st, slabs, (const f16*)bias, (f16*)out, ldo,
                           (int)M, (int)N, NP, tp.S);
        TGIS_CHECK_LAUNCH();
    }
    return TGIS_OK;
}

extern "C" int64_t tgis_gptq_gemm_fused_rows(int64_t K, int64_t groups, int act_order, int act) {
    if (K <= 0 || groups <= 0 || K % groups) return 0;
    static const int32_t some_perm = 0;
    return tall_ok(tall_min_m(), K, groups, act_order ? &some_perm : nullptr, act) ? std::max<int64_t>(256, tall_max_m()) : 256;
}

extern "C" int64_t tgis_gptq_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N) {
    GemmPlan pl = plan_gemm(K, N, 0, M);
    int64_t need = 4096 + slab_bytes(M, N, pl.S);
    if (M <= 64 && K % 64 == 0) need = std::max(need, 4096 + slab_bytes(M, N, gptq::wide_max_splits(K, N)));
    if (M >= tall_min_m() && K % 64 == 0) need = std::max(need, 4096 + tall_slab_bytes(M, N, plan_tall(M, K, N, 0).S));
    return need;
}

template <int TN, int WK, int ACT, bool G64, bool PERM, int MR>
static int launch_one(dim3 grid, size_t lds, hipStream_t st, const GemmArgs& a) {
    static bool attr_done = false;
    if (!attr_done) {
        TGIS_CHECK_HIP(hipFuncSetAttribute((const void*)gptq_gemm_kernel<TN, WK, ACT, G64, PERM, MR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 32 * RS * 2 + 64));
        attr_done = true;
    }
    hipLaunchKernelGGL((gptq_gemm_kernel<TN, WK, ACT, G64, PERM, MR>), grid, dim3(64 * TN * WK), lds, st, a);
    return TGIS_OK;
}
template <int TN, int WK, int ACT, bool G64, bool PERM>
static int launch_variant(int mr, dim3 grid, size_t lds, hipStream_t st, const GemmArgs& a) {
    if constexpr (WK == 2) {  // 64-row passes exist for two-k-part blocks only (LDS)
        if (mr == 2) return launch_one<TN, WK, ACT, G64, PERM, 2>(grid, lds, st, a);
    }
    return launch_one<TN, WK, ACT, G64, PERM, 1>(grid, lds, st, a);
}

static int64_t silu_split_below() {  // blocks of the unsplit plan under which a 64-row SiLU * up projection is split (0: never)
    static const int64_t v = getenv("TGIS_SILU_SPLIT_BELOW") ? atoll(getenv("TGIS_SILU_SPLIT_BELOW")) : 128;
    return v;
}

struct RopeEpi {
    const int32_t *positions, *slots;
    const f16 *cosb, *sinb;
    f16 *kpool, *vpool;
    int H, Hkv, D;
};

template <int CT, int ACT, bool OUTF, int MR>
static int launch_wide_mr(dim3 grid, hipStream_t st, const GemmArgs& a) {
    const size_t lds = gptq::wide_lds_bytes(CT);
    static bool attr_done = false;
    if (!attr_done) {
        TGIS_CHECK_HIP(hipFuncSetAttribute((const void*)gptq_wide_kernel<CT, ACT, OUTF, MR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((gptq_wide_kernel<CT, ACT, OUTF, MR>), grid, dim3(64 * gptq::WIDE_WK), lds, st, a);
    return TGIS_OK;
}
template <int CT, int ACT, bool OUTF>
static int launch_wide_one(dim3 grid, hipStream_t st, const GemmArgs& a) {
    return a.M > 32 ? launch_wide_mr<CT, ACT, OUTF, 2>(grid, st, a) : launch_wide_mr<CT, ACT, OUTF, 1>(grid, st, a);
}

static int launch_gptq(const void* x, int64_t ldx, const void* prepared, const void* bias, const int32_t* perm,
                       void* out, int64_t ldo, int64_t M, int64_t K, int64_t N, int64_t groups, int act, float* slabs,
                       int partial, const GemmPlan& pl, hipStream_t st, const RopeEpi* rope = nullptr) {
    PrepLayout p = prep_layout(K, N, groups);
    const int64_t mslabs = cdiv64(M, 32 * pl.MR);  // passes over the weights
    const int64_t gs = K / groups;
    const int64_t spg = gs / 64;  // k64-steps per group
    const bool group64 = groups == 1 || (gs % 64 == 0 && (spg & (spg - 1)) == 0);
    GemmArgs a;
    a.x = (const f16*)x;
    a.ldx = ldx;
    a.prep = (const uint8_t*)prepared;
    a.offB = p.offB;
    a.bias = (const f16*)bias;
    a.perm = perm;
    a.out = (f16*)out;
    a.ldo = ldo;
    a.M = (int)M;
    a.K = (int)K;
    a.N = (int)N;
    a.G = (int)groups;
    a.gs = (int)gs;
    a.KR = pl.KR;
    a.S = pl.S;
    a.NT = (int)p.NT;
    a.KS = (int)p.KS;
    a.slabs = slabs;
    a.partial = partial;
    a.spg_shift = 30;
    a.positions = a.slots = nullptr;
    a.cosb = a.sinb = nullptr;
    a.kpool = a.vpool = nullptr;
    a.rH = a.rHkv = a.rD = 0;
    if (rope) {
        a.positions = rope->positions;
        a.slots = rope->slots;
        a.cosb = rope->cosb;
        a.sinb = rope->sinb;
        a.kpool = rope->kpool;
        a.vpool = rope->vpool;
        a.rH = rope->H;
        a.rHkv = rope->Hkv;
        a.rD = rope->D;
    }
    if (groups > 1 && group64)
        for (a.spg_shift = 0; (1 << a.spg_shift) < spg; ++a.spg_shift) {}
    if (ldx == TGIS_LD_FRAGMENTS) {
        // activation in fragment order: the kernel of gptq_wide_body.h; `pl` carries its plan as TN = CT, S (callers use
        // wide_plan_as_gemm_plan); checked by the caller: wide_serves(), act in {0, 2, 3}, no permutation
        const bool outf = ldo == TGIS_LD_FRAGMENTS;
        dim3 wgrid((unsigned)cdiv64(p.NT, pl.TN), (unsigned)pl.S);
#define TGIS_WIDE(CT)                                                                                   \
    do {                                                                                                \
        int rc_ = act == 3   ? launch_wide_one<CT, 3, false>(wgrid, st, a)                              \
                  : act == 2 ? (outf ? launch_wide_one<CT, 2, true>(wgrid, st, a)                       \
                                     : launch_wide_one<CT, 2, false>(wgrid, st, a))                     \
                             : launch_wide_one<CT, 0, false>(wgrid, st, a);                             \
        if (rc_ != TGIS_OK) return rc_;                                                                 \
    } while (0)
        if (pl.TN == 4) TGIS_WIDE(4); else if (pl.TN == 3) TGIS_WIDE(3); else if (pl.TN == 1) TGIS_WIDE(1); else TGIS_WIDE(2);
#undef TGIS_WIDE
        TGIS_CHECK_LAUNCH();
        if (!partial && pl.S > 1) {
            const int NP = (int)p.NT * 32;
            dim3 rgrid((unsigned)cdiv64((int64_t)32 * (NP / 4), 256), (unsigned)cdiv64(M, 32));
            hipLaunchKernelGGL(splitk_reduce_f16_kernel, rgrid, dim3(256), 0, st, a.slabs, a.bias, a.out, a.ldo, a.M, a.N,
                               NP, a.S);
            TGIS_CHECK_LAUNCH();
        }
        return TGIS_OK;
    }
    dim3 grid((unsigned)cdiv64(p.NT, pl.TN), (unsigned)pl.S, (unsigned)mslabs);
    const size_t lds = (size_t)pl.WK * 2 * 32 * pl.MR * RS * sizeof(f16) + 64;  // x buffers + arrival counters
#define TGIS_LAUNCH_GEMM(T, W, A, G, P)                                                        \
    do {                                                                                       \
        int rc_ = launch_variant<T, W, A, G, P>(pl.MR, grid, lds, st, a);                      \
        if (rc_ != TGIS_OK) return rc_;                                                        \
    } while (0)
#define TGIS_LAUNCH_GEMM_W(A, G, P)                      \
    do {                                                 \
        const int tw = pl.TN * 10 + pl.WK;               \
        if (tw == 44)                                    \
            TGIS_LAUNCH_GEMM(4, 4, A, G, P);             \
        else if (tw == 42)                               \
            TGIS_LAUNCH_GEMM(4, 2, A, G, P);             \
        else if (tw == 34)                               \
            TGIS_LAUNCH_GEMM(3, 4, A, G, P);             \
        else if (tw == 32)                               \
            TGIS_LAUNCH_GEMM(3, 2, A, G, P);             \
        else if (tw == 24)                               \
            TGIS_LAUNCH_GEMM(2, 4, A, G, P);             \
        else                                             \
            TGIS_LAUNCH_GEMM(2, 2, A, G, P);             \
    } while (0)
    if (act == 3) {  // rope epilogue: 64 * 2^n groups, no act-order (checked by the caller)
        TGIS_LAUNCH_GEMM_W(3, true, false);
        TGIS_CHECK_LAUNCH();
        return TGIS_OK;
    }
    const int variant = (act == 1 ? 4 : act == 2 ? 8 : 0) | (group64 ? 2 : 0) | (perm ? 1 : 0);
    switch (variant) {
        case 8: TGIS_LAUNCH_GEMM_W(2, false, false); break;
        case 9: TGIS_LAUNCH_GEMM_W(2, false, true); break;
        case 10: TGIS_LAUNCH_GEMM_W(2, true, false); break;
        case 11: TGIS_LAUNCH_GEMM_W(2, true, true); break;
        case 0: TGIS_LAUNCH_GEMM_W(0, false, false); break;
        case 1: TGIS_LAUNCH_GEMM_W(0, false, true); break;
        case 2: TGIS_LAUNCH_GEMM_W(0, true, false); break;
        case 3: TGIS_LAUNCH_GEMM_W(0, true, true); break;
        case 4: TGIS_LAUNCH_GEMM_W(1, false, false); break;
        case 5: TGIS_LAUNCH_GEMM_W(1, false, true); break;
        case 6: TGIS_LAUNCH_GEMM_W(1, true, false); break;
        case 7: TGIS_LAUNCH_GEMM_W(1, true, true); break;
    }
#undef TGIS_LAUNCH_GEMM_W
#undef TGIS_LAUNCH_GEMM
    TGIS_CHECK_LAUNCH();
    if (!partial && pl.S > 1 && !getenv("TGIS_GPTQ_NOREDUCE")) {
        const int NP = (int)p.NT * 32;
        dim3 rgrid((unsigned)cdiv64((int64_t)32 * (NP / 4), 256), (unsigned)cdiv64(M, 32));
        hipLaunchKernelGGL(splitk_reduce_f16_kernel, rgrid, dim3(256), 0, st, a.slabs, a.bias, a.out, a.ldo, a.M, a.N,
                           NP, a.S);
        TGIS_CHECK_LAUNCH();
    }
    return TGIS_OK;
}

static int check_gemm_args(const void* x, int64_t ldx, const void* prepared, int64_t M, int64_t K, int64_t N,
                           int64_t groups, int act) {
    TGIS_CHECK_ARG(x && prepared, "tgis_gptq_gemm: null tensor");
    TGIS_CHECK_ARG(M >= 0 && K > 0 && N > 0 && K % 32 == 0 && N % 32 == 0, "tgis_gptq_gemm: bad shape");
    TGIS_CHECK_ARG(groups > 0 && K % groups == 0, "tgis_gptq_gemm: K %% groups != 0");
    TGIS_CHECK_ARG(act == 0 || act == 1 || act == 2, "tgis_gptq_gemm: act must be 0, 1 or 2");
    TGIS_CHECK_ARG(act != 2 || N % 32 == 0, "tgis_gptq_gemm: act=2 needs N/2 to be a multiple of 16");
    if (ldx == TGIS_LD_FRAGMENTS) {
        TGIS_CHECK_ARG(((uintptr_t)x % 16) == 0 && gptq::wide_serves(M, K, N, groups, false) && act != 1,
                       "tgis_gptq_gemm: an activation in fragment order needs 1 <= M <= 64, K %% 64 == 0, groups of 64 * 2^n "
                       "rows and act 0 or 2 (M=%ld K=%ld N=%ld groups=%ld act=%d)", (long)M, (long)K, (long)N, (long)groups, act);
        return TGIS_OK;
    }
    TGIS_CHECK_ARG(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0, "tgis_gptq_gemm: x must be 16-byte aligned rows");
    return TGIS_OK;
}

// the fragment-order kernel's plan in the GemmPlan that launch_gptq takes (TN = column tiles per wave, S = k splits)
static GemmPlan wide_plan_as_gemm_plan(int64_t K, int64_t N, int act, int64_t M, bool finished = false) {
    const gptq::WidePlan w = gptq::plan_wide(K, N, act, M, finished);
    return {0, w.S, gptq::WIDE_WK, w.CT, 1};
}

extern "C" int tgis_gptq_gemm_f16(const void* x, int64_t ldx, const void* prepared, const void* bias,
                                  const int32_t* perm, void* out, int64_t ldo, int64_t M, int64_t K,
                                  int64_t N, int64_t groups, int act, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    int rc = check_gemm_args(x, ldx, prepared, M, K, N, groups, act);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(out, "tgis_gptq_gemm_f16: null out");
    if (M == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    if (ldx == TGIS_LD_FRAGMENTS) {
        TGIS_CHECK_ARG(!perm, "tgis_gptq_gemm_f16: act-order matrices take a row-major activation");
        TGIS_CHECK_ARG(ldo != TGIS_LD_FRAGMENTS || (act == 2 && (N / 2) % 64 == 0),
                       "tgis_gptq_gemm_f16: only the act = 2 output (N / 2 a multiple of 64) can leave in fragment order");
        const GemmPlan wp = wide_plan_as_gemm_plan(K, N, act, M, /*finished=*/true);  // this entry point returns f16
        const int64_t need_w = 4096 + slab_bytes(M, N, wp.S);
        TGIS_CHECK_ARG(workspace && workspace_bytes >= need_w, "tgis_gptq_gemm_f16: workspace too small (%ld < %ld)",
                       (long)workspace_bytes, (long)need_w);
        TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
        return launch_gptq(x, ldx, prepared, bias, nullptr, out, ldo, M, K, N, groups, act,
                           (float*)((uint8_t*)workspace + 4096), 0, wp, st);
    }
    TGIS_CHECK_ARG(ldo != TGIS_LD_FRAGMENTS, "tgis_gptq_gemm_f16: a fragment-order output needs a fragment-order activation");
    if (tall_ok(M, K, groups, perm, act)) {
        const TallPlan tp = plan_tall(M, K, N, act);
        const int64_t need_t = 4096 + (tp.S > 1 ? tall_slab_bytes(M, N, tp.S) : 0);
        TGIS_CHECK_ARG(workspace && workspace_bytes >= need_t, "tgis_gptq_gemm_f16: workspace too small (%ld < %ld)",
                       (long)workspace_bytes, (long)need_t);
        TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
        return launch_tall(x, ldx, prepared, bias, out, ldo, M, K, N, groups, act, (float*)((uint8_t*)workspace + 4096), 0,
                           tp, st);
    }
    GemmPlan pl = plan_gemm(K, N, act, M);
    TGIS_CHECK_ARG(cdiv64(M, 32) <= 65535, "tgis_gptq_gemm_f16: M too large for one launch");
    // SiLU * up whose unsplit plan has few blocks (TP shards): every block would take in its whole M x K activation
    // through one CU (2-8 x its weights: a 70B gate_up shard at TP = 8 and 64 rows, 112 blocks, 34.6 us).  The projection
    // then runs with the split plan of a plain GEMM and the activation moves into the split-K reduce (us, fused -> split:
    // 64 rows 8192x7168 34.2 -> 24.0, 4096x2752 18.7 -> 11.3; 32 rows 8192x7168 19.0 -> 15.5, 4096x2752 10.8 -> 8.6,
    // 4096x5504 (86 blocks) 11.2 -> 10.3; from 172 blocks on the epilogue wins: 4096x11008 11.8 vs 17.5).
    bool split_silu = false;
    if (act == 2 && cdiv64(cdiv64(N, 32), pl.TN) < silu_split_below()) {
        const GemmPlan ps = plan_gemm(K, N, 0, M);
        if (ps.S > 1) {
            pl = ps;
            split_silu = true;
        }
    }
    int64_t need = 4096 + slab_bytes(M, N, pl.S);
    TGIS_CHECK_ARG(workspace && workspace_bytes >= need, "tgis_gptq_gemm_f16: workspace too small (%ld < %ld)",
                   (long)workspace_bytes, (long)need);
    TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
    if (split_silu) {
        float* slabs = (float*)((uint8_t*)workspace + 4096);
        rc = launch_gptq(x, ldx, prepared, nullptr, perm, out, ldo, M, K, N, groups, 0, slabs, 1, pl, st);
        if (rc != TGIS_OK) return rc;
        const int NP = (int)cdiv64(N, 32) * 32;
        dim3 rgrid((unsigned)cdiv64((int64_t)32 * (NP / 32) * 4, 256), (unsigned)cdiv64(M, 32));
        hipLaunchKernelGGL(splitk_reduce_silu_kernel, rgrid, dim3(256), 0, st, slabs, (const f16*)bias, (f16*)out, ldo, (int)M,
                           (int)N, NP, pl.S);
        TGIS_CHECK_LAUNCH();
        return TGIS_OK;
    }
    return launch_gptq(x, ldx, prepared, bias, perm, out, ldo, M, K, N, groups, act,
                       (float*)((uint8_t*)workspace + 4096), 0, pl, st);
}

// ---- qkv projection with the rotary embedding and the cache write in its epilogue ------------------------------------
static int64_t rope_min_blocks(int64_t M) {
    static const int64_t min_blocks = getenv("TGIS_ROPE_MIN_BLOCKS") ? atoll(getenv("TGIS_ROPE_MIN_BLOCKS")) : 128;
    return M <= 32 ? std::min<int64_t>(min_blocks, 48) : min_blocks;
}

extern "C" int tgis_gptq_rope_ok(int64_t M, int64_t K, int64_t N, int64_t groups, int act_order, int64_t D) {
    if (M < 1 || M > 64 || act_order || groups <= 0 || K % groups || D < 32 || D % 32 || N <= 0 || N % D) return 0;
    const int64_t gs = K / groups, spg = gs / 64;
    if (!(groups == 1 || (gs % 64 == 0 && (spg & (spg - 1)) == 0))) return 0;
    // The epilogue needs the whole k range in one block (no split-K): worth it only while that plan still covers the chip
    // (measured: 7B qkv at 32 rows, 192 blocks: 14.9 vs 17.2 us for the pair; 70B qkv at 64 rows, 80 blocks of 42 MB: +5 % on
    // the step; a TP = 8 shard of the 7B qkv, 24 blocks: +2 % on the rank-step; TinyLlama dense, 40 blocks: +-0).
    // Round 6: counted in blocks of the kernel that will run it (the fragment-order kernel gives a wave ONE tile where two would
    // leave fewer than 128 blocks), and up to 32 rows it pays from 48 blocks on — one rank of cfg3 at TP 8 / 4 / 2 (48 / 96 / 192
    // one-tile blocks) 1.838 / 2.110 / 3.000 -> 1.815 / 2.072 / 2.903 ms per step; at 64 rows every block takes in twice the
    // activation and the bar stays at 128 (a cfg4 rank at TP 8, 40 blocks: 6.45 -> 7.10 ms).  profiles/r06_tp_rope_blocks.log
    const GemmPlan pl = plan_gemm(K, N, 2, M);
    const int64_t blocks = gptq::wide_serves(M, K, N, groups, false) ? gptq::wide_blocks(K, N, 3, M)
                                                                     : cdiv64(cdiv64(N, 32), pl.TN);
    return blocks >= rope_min_blocks(M) ? 1 : 0;
}

extern "C" int tgis_gptq_gemm_rope_f16(const void* x, int64_t ldx, const void* prepared, const void* bias,
                                       const int32_t* positions, const int32_t* slots, const void* cos, const void* sin,
                                       void* q_out, int64_t ldq, void* k_pool, void* v_pool, int64_t M, int64_t K, int64_t N,
                                       int64_t groups, int64_t H, int64_t Hkv, int64_t D, void* stream) {
    int rc = check_gemm_args(x, ldx, prepared, M, K, N, groups, 0);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(positions && slots && cos && sin && q_out && k_pool && v_pool, "tgis_gptq_gemm_rope_f16: null tensor");
    {
        const int64_t gs = groups > 0 && K % groups == 0 ? K / groups : 0, spg = gs / 64;
        TGIS_CHECK_ARG(M >= 1 && M <= 64 && D >= 32 && D % 32 == 0 &&
                           (groups == 1 || (gs > 0 && gs % 64 == 0 && (spg & (spg - 1)) == 0)),
                       "tgis_gptq_gemm_rope_f16: needs 1 <= M <= 64, groups of 64 * 2^n rows and a head size that is a "
                       "multiple of 32 (M=%ld K=%ld groups=%ld D=%ld)", (long)M, (long)K, (long)groups, (long)D);
    }
    TGIS_CHECK_ARG(H >= 1 && Hkv >= 1 && (H + 2 * Hkv) * D == N && ldq >= H * D,
                   "tgis_gptq_gemm_rope_f16: N must be (H + 2 Hkv) * D and q rows must hold H * D elements");
    GemmPlan pl = plan_gemm(K, N, 2, M);  // as the SiLU epilogue: the whole k range in one block (S == 1)
    if (ldx == TGIS_LD_FRAGMENTS) pl = wide_plan_as_gemm_plan(K, N, 3, M);
    RopeEpi rope{positions, slots, (const f16*)cos, (const f16*)sin, (f16*)k_pool, (f16*)v_pool, (int)H, (int)Hkv, (int)D};
    hipStream_t st = (hipStream_t)stream;
    TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
    return launch_gptq(x, ldx, prepared, bias, nullptr, q_out, ldq, M, K, N, groups, 3, nullptr, 0, pl, st, &rope);
}


extern "C" int64_t tgis_gptq_gemm_partial_bytes(int64_t M, int64_t K, int64_t N) {
    GemmPlan pl = plan_gemm(K, N, 0, M);
    int64_t need = cdiv64(std::max<int64_t>(M, 1), 64) * 2 * pl.S * 32 * cdiv64(N, 32) * 32 * 4;
    if (M <= 64 && K % 64 == 0)  // the fragment-order kernel's plan may split further
        need = std::max<int64_t>(need, (int64_t)2 * gptq::wide_max_splits(K, N) * 32 * cdiv64(N, 32) * 32 * 4);
    if (M >= tall_min_m() && K % 64 == 0) need = std::max(need, tall_slab_bytes(M, N, plan_tall(M, K, N, 0).S));
    return need;
}

// Is an activation in fragment order (TGIS_LD_FRAGMENTS) served for this GEMM, and expected to beat the row-major launch?
// act 2 / 3 keep the whole k range in a block: only while that still covers the chip (>= 128 blocks, as tgis_gptq_rope_ok).
extern "C" int tgis_gptq_fragments_ok(int64_t M, int64_t K, int64_t N, int64_t groups, int act_order, int act) {
    if (groups <= 0 || K <= 0 || N <= 0 || !gptq::wide_serves(M, K, N, groups, act_order != 0)) return 0;
    if (act != 0 && act != 2 && act != 3) return 0;
    static const bool off = getenv("TGIS_GPTQ_FRAGMENTS") && atoi(getenv("TGIS_GPTQ_FRAGMENTS")) == 0;
    if (off) return 0;
    static const int64_t max_rows = getenv("TGIS_GPTQ_FRAGMENTS_MAX_ROWS") ? atoll(getenv("TGIS_GPTQ_FRAGMENTS_MAX_ROWS")) : 64;
    if (M > max_rows) return 0;
    // SiLU * up: from 64 blocks on (round 5, a 7B gate_up shard at TP = 8, 86 one-tile blocks: 8.2 us against 6.3 + 4.7 for
    // the split streaming kernel + its reduce, profiles/r05_tp8_variants.log)
    static const int64_t min_blocks_silu = getenv("TGIS_SILU_MIN_BLOCKS") ? atoll(getenv("TGIS_SILU_MIN_BLOCKS")) : 64;
    if (act == 3 && gptq::wide_blocks(K, N, act, M) < rope_min_blocks(M)) return 0;
    if (act == 2 && gptq::wide_blocks(K, N, act, M, true) < min_blocks_silu) return 0;
    return 1;
}

extern "C" int tgis_gptq_gemm_f16_partial(const void* x, int64_t ldx, const void* prepared, const int32_t* perm,
                                          int64_t M, int64_t K, int64_t N, int64_t groups, int act, float* slabs,
                                          int64_t slabs_bytes, int* num_slabs, int64_t* slab_ld, void* stream) {
    int rc = check_gemm_args(x, ldx, prepared, M, K, N, groups, act);
    if (rc != TGIS_OK) return rc;
    TGIS_CHECK_ARG(M >= 1 && cdiv64(M, 32) <= 65535, "tgis_gptq_gemm_f16_partial: bad M");
    TGIS_CHECK_ARG(act != 2, "tgis_gptq_gemm_f16_partial: act=2 cannot be deferred");
    if (tall_ok(M, K, groups, perm, act)) {
        const TallPlan tp = plan_tall(M, K, N, act);
        TGIS_CHECK_ARG(slabs && slabs_bytes >= tall_slab_bytes(M, N, tp.S),
                       "tgis_gptq_gemm_f16_partial: slab buffer too small");
        if (num_slabs) *num_slabs = tp.S;
        if (slab_ld) *slab_ld = cdiv64(N, 32) * 32;
        TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, (hipStream_t)stream);
        return launch_tall(x, ldx, prepared, nullptr, nullptr, 0, M, K, N, groups, act, slabs, 1, tp, (hipStream_t)stream);
    }
    GemmPlan pl = plan_gemm(K, N, 0, M);
    if (ldx == TGIS_LD_FRAGMENTS) {
        TGIS_CHECK_ARG(!perm && act == 0, "tgis_gptq_gemm_f16_partial: fragment-order activations: act 0, no act-order");
        pl = wide_plan_as_gemm_plan(K, N, 0, M);
    }
    TGIS_CHECK_ARG(slabs && slabs_bytes >= tgis_gptq_gemm_partial_bytes(M, K, N),
                   "tgis_gptq_gemm_f16_partial: slab buffer too small");
    hipStream_t st = (hipStream_t)stream;
    if (num_slabs) *num_slabs = pl.S;
    if (slab_ld) *slab_ld = cdiv64(N, 32) * 32;
    TgisTimedScope timed(TGIS_OP_GPTQ_GEMM, st);
    return launch_gptq(x, ldx, prepared, nullptr, perm, nullptr, 0, M, K, N, groups, act, slabs, 1, pl, st);
}

// debug aid (not part of the documented ABI): resident blocks per CU the runtime reports for the main kernel
#ifdef TGIS_TRACE
extern "C" int tgis_debug_set_trace(void* ptr) {
    long long* p = (long long*)ptr;
    TGIS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gptq::g_trace), &p, sizeof(p)));
    return TGIS_OK;
}
#endif

extern "C" int tgis_debug_gemm_occupancy(int tn, int wk) {
    int nb = -1;
    const size_t lds = (size_t)wk * 2 * 32 * RS * sizeof(f16) + 64;
    if (tn == 4 && wk == 4)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gptq_gemm_kernel<4, 4, 0, true, false, 1>, 1024, lds);
    else if (tn == 2 && wk == 4)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gptq_gemm_kernel<2, 4, 0, true, false, 1>, 512, lds);
    else if (tn == 2 && wk == 2)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gptq_gemm_kernel<2, 2, 0, true, false, 1>, 256, lds);
    else if (tn == 3 && wk == 4)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gptq_gemm_kernel<3, 4, 0, true, false, 1>, 768, lds);
    return nb;
}

extern "C" int tgis_gptq_dequant_f16(const void* prepared, void* w_out, int64_t K, int64_t N, int64_t groups,
                                     int flags, void* stream) {
    TGIS_CHECK_ARG(prepared && w_out && K > 0 && N > 0 && groups > 0 && K % groups == 0,
                   "tgis_gptq_dequant_f16: bad arguments");
    PrepLayout p = prep_layout(K, N, groups);
    int64_t total = p.NT * p.KS * 256;
    hipLaunchKernelGGL(gptq_dequant_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)prepared, p.offB, (f16*)w_out, (int)K, (int)N, (int)groups,
                       (int)(K / groups), (int)p.NT, (int)p.KS, flags);
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}
