// Fused residual-add + RMSNorm / LayerNorm for gfx950.
// Replaces dropout_layer_norm.dropout_add_ln_fwd as called with p=0 from
// custom_modeling/flash_llama_modeling.py:132-152 (RMSNorm) and utils/layers.py:376-396 (LayerNorm):
//   res = x (+ residual)  [sum formed in fp32, stored in the model dtype]
//   y   = norm(res) * weight (+ bias)   [statistics in fp32 from the fp32 sum, one rounding at the end]
// One 256-thread workgroup per row, 16-byte loads, row cached in registers (hidden <= 16384).
#include "common.h"

namespace {

constexpr int MAX_HIDDEN = 16384;

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) t += sh[k];
    return t;
}

// x either as a model-dtype tensor, or (PARTIAL) as S fp32 split-K slabs [S][32][slab_ld] whose sum (+xbias) is
// rounded to the model dtype first — bit-identical to running the split-K reduce kernel and then this one.
// NT threads per row: 256 in general, 512 for decode-sized batches (few rows: latency matters, not occupancy)
template <typename T, bool RMS, bool PARTIAL, int NT>
__global__ __launch_bounds__(NT) void norm_kernel(const T* x, const T* residual,
                                                  const T* __restrict__ weight, const T* __restrict__ bias,
                                                  T* y, T* res_out, int hidden,
                                                  float eps, const float* __restrict__ slabs, int S, int64_t slab_ld,
                                                  const T* __restrict__ xbias, int y_frag) {
    using V8 = typename VecT<T>::x8;
    constexpr int MAXV = MAX_HIDDEN / (NT * 8);
    __shared__ float sh[NT / 64];
    const int64_t row = blockIdx.x;
    const T* xr = PARTIAL ? nullptr : x + row * hidden;
    const T* rr = residual ? residual + row * hidden : nullptr;
    float v[MAXV][8];
    V8 wvs[MAXV], bvs[MAXV];  // weight / bias: asked for with the inputs, not after the row reductions
    const int nchunk = hidden >> 3;  // hidden % 8 == 0
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
        int c = threadIdx.x + it * NT;
        if (c < nchunk) {
            wvs[it] = ld16<V8>(weight + c * 8);
            if (bias) bvs[it] = ld16<V8>(bias + c * 8);
            // Round 5: the residual and the bias of the GEMM before are requested HERE, in front of the slabs.  Behind them (in
            // their own basic blocks, after the waits of the slab sum) they were a second and a third dependent round trip to
            // memory in a kernel that is nothing but one round trip.
            V8 b, bv;
            if (rr) b = ld16<V8>(rr + c * 8);
            if (PARTIAL && xbias) bv = ld16<V8>(xbias + c * 8);
            V8 a;
            if (PARTIAL) {
                f32x4 lo, hi;
                // slabs are stored in 32-row units: [row / 32][S][32][slab_ld]
                sum_slabs8(slabs + ((int64_t)(row >> 5) * S * 32 + (row & 31)) * slab_ld + c * 8, 32 * slab_ld, S, lo, hi);
                if (xbias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] += to_f32(bv[e]);
                        hi[e] += to_f32(bv[e + 4]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = from_f32<T>(lo[e]);
                    a[e + 4] = from_f32<T>(hi[e]);
                }
            } else {
                a = ld16<V8>(xr + c * 8);
            }
            V8 o;
            if (rr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = to_f32(a[e]) + to_f32(b[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = to_f32(a[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = from_f32<T>(v[it][e]);
                s1 += v[it][e];
                s2 += v[it][e] * v[it][e];
            }
            if (res_out) st16(res_out + row * hidden + c * 8, o);
        }
    }
    float mean = 0.f, rstd;
    if (RMS) {
        float tot = block_sum<NT>(s2, sh);
        rstd = rsqrtf(tot / hidden + eps);
    } else {
        mean = block_sum<NT>(s1, sh) / hidden;
        float d2 = 0.f;
#pragma unroll
        for (int it = 0; it < MAXV; ++it) {
            int c = threadIdx.x + it * NT;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float d = v[it][e] - mean;
                    d2 += d * d;
                }
            }
        }
        float var = block_sum<NT>(d2, sh) / hidden;
        rstd = rsqrtf(var + eps);
    }
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
        int c = threadIdx.x + it * NT;
        if (c < nchunk) {
            const V8 wv = wvs[it];
            V8 o;
            if (bias) {
                const V8 bv = bvs[it];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = from_f32<T>((v[it][e] - mean) * rstd * to_f32(wv[e]) + to_f32(bv[e]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = from_f32<T>((v[it][e] - mean) * rstd * to_f32(wv[e]));
            }
            // y_frag (RMSNorm in front of an int4 decode GEMM): 32-row fragment order, xf_off in common.h
            st16(y + (y_frag ? xf_off(row, c * 8, hidden) : row * hidden + c * 8), o);
        }
    }
}

template <bool RMS>
static int launch_norm(const void* x, const void* residual, const void* weight, const void* bias, void* y,
                       void* res_out, int64_t rows, int64_t hidden, float eps, int dtype, void* stream,
                       const float* slabs = nullptr, int S = 0, int64_t slab_ld = 0, const void* xbias = nullptr,
                       int64_t ldy = 0) {
    TGIS_CHECK_ARG((x || slabs) && weight && y, "norm: null tensor");
    TGIS_CHECK_ARG(!slabs || (S >= 1 && slab_ld >= hidden && slab_ld % 4 == 0),
                   "norm: partial input needs a slab row stride >= hidden");
    TGIS_CHECK_ARG(hidden > 0 && hidden % 8 == 0 && hidden <= MAX_HIDDEN,
                   "norm: hidden (%ld) must be a multiple of 8 and <= %d", (long)hidden, MAX_HIDDEN);
    TGIS_CHECK_ARG(dtype == TGIS_F16 || dtype == TGIS_BF16, "norm: bad dtype");
    const int y_frag = ldy == TGIS_LD_FRAGMENTS;
    TGIS_CHECK_ARG(ldy == 0 || ldy == hidden || (y_frag && RMS && rows <= 64 && hidden % 64 == 0),
                   "norm: y is [rows, hidden] contiguous (ldy = 0 or hidden), or — RMSNorm, rows <= 64, hidden %% 64 == 0 — in "
                   "fragment order (ldy = TGIS_LD_FRAGMENTS)");
    if (rows == 0) return TGIS_OK;
    hipStream_t st = (hipStream_t)stream;
    TgisTimedScope timed(TGIS_OP_NORM, st);
    const bool wide = rows <= 64 && hidden >= 2048;
#define TGIS_NORM_LAUNCH(T, P)                                                                                   \
    do {                                                                                                         \
        if (wide)                                                                                                \
            hipLaunchKernelGGL((norm_kernel<T, RMS, P, 512>), dim3((unsigned)rows), dim3(512), 0, st, (const T*)x, \
                               (const T*)residual, (const T*)weight, (const T*)bias, (T*)y, (T*)res_out,         \
                               (int)hidden, eps, slabs, S, slab_ld, (const T*)xbias, y_frag);                    \
        else                                                                                                     \
            hipLaunchKernelGGL((norm_kernel<T, RMS, P, 256>), dim3((unsigned)rows), dim3(256), 0, st, (const T*)x, \
                               (const T*)residual, (const T*)weight, (const T*)bias, (T*)y, (T*)res_out,         \
                               (int)hidden, eps, slabs, S, slab_ld, (const T*)xbias, y_frag);                    \
    } while (0)
    if (dtype == TGIS_F16) {
        if (slabs) TGIS_NORM_LAUNCH(f16, true); else TGIS_NORM_LAUNCH(f16, false);
    } else {
        if (slabs) TGIS_NORM_LAUNCH(bf16, true); else TGIS_NORM_LAUNCH(bf16, false);
    }
#undef TGIS_NORM_LAUNCH
    TGIS_CHECK_LAUNCH();
    return TGIS_OK;
}

}  // namespace

extern "C" int tgis_rmsnorm_residual(const void* x, const void* residual, const void* weight, void* y, int64_t ldy,
                                     void* res_out, int64_t rows, int64_t hidden, float eps, int dtype,
                                     void* stream) {
    return launch_norm<true>(x, residual, weight, nullptr, y, res_out, rows, hidden, eps, dtype, stream, nullptr, 0, 0,
                             nullptr, ldy);
}

extern "C" int tgis_rmsnorm_residual_partial(const float* slabs, int num_slabs, int64_t slab_ld, const void* bias,
                                             const void* residual, const void* weight, void* y, int64_t ldy, void* res_out,
                                             int64_t rows, int64_t hidden, float eps, int dtype, void* stream) {
    TGIS_CHECK_ARG(slabs, "tgis_rmsnorm_residual_partial: null slabs");
    return launch_norm<true>(nullptr, residual, weight, nullptr, y, res_out, rows, hidden, eps, dtype, stream, slabs,
                             num_slabs, slab_ld, bias, ldy);
}

extern "C" int tgis_layernorm_residual_partial(const float* slabs, int num_slabs, int64_t slab_ld, const void* xbias,
                                               const void* residual, const void* weight, const void* bias, void* y,
                                               void* res_out, int64_t rows, int64_t hidden, float eps, int dtype,
                                               void* stream) {
    TGIS_CHECK_ARG(slabs, "tgis_layernorm_residual_partial: null slabs");
    return launch_norm<false>(nullptr, residual, weight, bias, y, res_out, rows, hidden, eps, dtype, stream, slabs,
                              num_slabs, slab_ld, xbias);
}

extern "C" int tgis_layernorm_residual(const void* x, const void* residual, const void* weight,
                                       const void* bias, void* y, void* res_out, int64_t rows, int64_t hidden,
                                       float eps, int dtype, void* stream) {
    return launch_norm<false>(x, residual, weight, bias, y, res_out, rows, hidden, eps, dtype, stream);
}
