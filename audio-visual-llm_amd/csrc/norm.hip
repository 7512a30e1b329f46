// LayerNorm / RMSNorm for gfx950.  HBM-bound: one pass over x (row kept in registers), 8/16-byte vector
// accesses, wave-shuffle reductions (one wave per row up to d=2048, one 256-thread block per row above).
// Reference arithmetic: nn.LayerNorm as used by HF whisper (:379-413) / clip (:362-384) encoder layers and
// LlamaRMSNorm (HF:models/llama/modeling_llama.py:62-67: fp32 variance, eps inside rsqrt).
#include "common.h"
#include "avllm_internal.h"

namespace {

constexpr int MAXV = 8;   // vectors of 4 per thread

template <int NT> __device__ __forceinline__ float row_sum(float v, float* red) {
    if constexpr (NT == 64) return wave_sum(v);
    else return block_sum(v, red);
}

// MODE 0: layernorm (w,b)   MODE 1: rmsnorm fwd (w, writes rstd)
template <typename T, int NT, int MODE>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                       const T* __restrict__ b, T* __restrict__ y,
                                                       float* __restrict__ rstd_out, long rows, int d, float eps) {
    __shared__ float red[8];
    constexpr int RPB = 256 / NT;
    const int sub = threadIdx.x / NT, t = threadIdx.x % NT;
    const long row = (long)blockIdx.x * RPB + sub;
    const bool active = row < rows;           // NT==256: uniform; NT==64: per wave
    if (NT == 64 && !active) return;
    const T* xr = x + (active ? row : 0) * d;
    float v[MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * NT + t) * 4;
        if (c < d) { load_f<4>(xr + c, v[i]); s += v[i][0] + v[i][1] + v[i][2] + v[i][3]; }
        else { v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f; }
    }
    float mean = 0.f;
    if (MODE == 0) mean = row_sum<NT>(s, red) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * NT + t) * 4;
        if (c < d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float u = v[i][j] - mean; q += u * u; }
        }
    }
    const float var = row_sum<NT>(q, red) / d;
    const float rstd = rsqrtf(var + eps);
    if (MODE == 1 && rstd_out && t == 0 && active) rstd_out[row] = rstd;
    if (!active) return;
    T* yr = y + row * d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * NT + t) * 4;
        if (c < d) {
            float wv[4], o[4];
            load_f<4>(w + c, wv);
            if (MODE == 0) {
                float bv[4];
                load_f<4>(b + c, bv);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = wv[j] * (v[i][j] * rstd);
            }
            store_f<4>(yr + c, o);
        }
    }
}

// dx = dres + rstd*(w*dy) - x*rstd^3*mean(w*dy*x)
template <typename T, int NT>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const T* __restrict__ w, const float* __restrict__ rstd,
                                                          const T* __restrict__ dres, T* __restrict__ dx, long rows, int d) {
    __shared__ float red[8];
    constexpr int RPB = 256 / NT;
    const int sub = threadIdx.x / NT, t = threadIdx.x % NT;
    const long row = (long)blockIdx.x * RPB + sub;
    const bool active = row < rows;
    if (NT == 64 && !active) return;
    const long r = active ? row : 0;
    float xv[MAXV][4], gv[MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * NT + t) * 4;
        if (c < d) {
            float wv[4], dv[4];
            load_f<4>(x + r * d + c, xv[i]);
            load_f<4>(dy + r * d + c, dv);
            load_f<4>(w + c, wv);
#pragma unroll
            for (int j = 0; j < 4; ++j) { gv[i][j] = wv[j] * dv[j]; s += gv[i][j] * xv[i][j]; }
        }
    }
    const float dot = row_sum<NT>(s, red);
    if (!active) return;
    const float rs = rstd[row];
    const float coef = dot * rs * rs * rs / d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * NT + t) * 4;
        if (c < d) {
            float o[4], dr[4] = {0.f, 0.f, 0.f, 0.f};
            if (dres) load_f<4>(dres + row * d + c, dr);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = dr[j] + rs * gv[i][j] - xv[i][j] * coef;
            store_f<4>(dx + row * d + c, o);
        }
    }
}

template <typename T, int MODE>
int launch_fwd(const void* x, const void* w, const void* b, void* y, float* rstd, long rows, int d, float eps, hipStream_t st) {
    if (d <= 64 * 4 * MAXV) {
        hipLaunchKernelGGL((norm_fwd_kernel<T, 64, MODE>), dim3(av_cdiv(rows, 4)), dim3(256), 0, st, (const T*)x, (const T*)w,
                           (const T*)b, (T*)y, rstd, rows, d, eps);
    } else {
        hipLaunchKernelGGL((norm_fwd_kernel<T, 256, MODE>), dim3(rows), dim3(256), 0, st, (const T*)x, (const T*)w,
                           (const T*)b, (T*)y, rstd, rows, d, eps);
    }
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // namespace

int av_layernorm(const void* x, const void* w, const void* b, void* y, long rows, int d, float eps, int dtype, hipStream_t st) {
    AV_CHECK_ARG(x && w && b && y && rows > 0, "layernorm: null/empty");
    AV_CHECK_ARG(d % 4 == 0 && d <= 256 * 4 * MAXV, "layernorm: d=%d unsupported", d);
    return dtype == AV_F32 ? launch_fwd<float, 0>(x, w, b, y, nullptr, rows, d, eps, st)
                           : launch_fwd<bf16, 0>(x, w, b, y, nullptr, rows, d, eps, st);
}

int av_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, long rows, int d, float eps, int dtype, hipStream_t st) {
    AV_CHECK_ARG(x && w && y && rows > 0, "rmsnorm: null/empty");
    AV_CHECK_ARG(d % 4 == 0 && d <= 256 * 4 * MAXV, "rmsnorm: d=%d unsupported", d);
    return dtype == AV_F32 ? launch_fwd<float, 1>(x, w, nullptr, y, rstd, rows, d, eps, st)
                           : launch_fwd<bf16, 1>(x, w, nullptr, y, rstd, rows, d, eps, st);
}

int av_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres_in, void* dx_out,
                   long rows, int d, int dtype, hipStream_t st) {
    AV_CHECK_ARG(dy && x && w && rstd && dx_out && rows > 0, "rmsnorm_bwd: null/empty");
    AV_CHECK_ARG(d % 4 == 0 && d <= 256 * 4 * MAXV, "rmsnorm_bwd: d=%d unsupported", d);
    if (dtype == AV_F32) {
        if (d <= 64 * 4 * MAXV)
            hipLaunchKernelGGL((rmsnorm_bwd_kernel<float, 64>), dim3(av_cdiv(rows, 4)), dim3(256), 0, st, (const float*)dy,
                               (const float*)x, (const float*)w, rstd, (const float*)dres_in, (float*)dx_out, rows, d);
        else
            hipLaunchKernelGGL((rmsnorm_bwd_kernel<float, 256>), dim3(rows), dim3(256), 0, st, (const float*)dy,
                               (const float*)x, (const float*)w, rstd, (const float*)dres_in, (float*)dx_out, rows, d);
    } else {
        if (d <= 64 * 4 * MAXV)
            hipLaunchKernelGGL((rmsnorm_bwd_kernel<bf16, 64>), dim3(av_cdiv(rows, 4)), dim3(256), 0, st, (const bf16*)dy,
                               (const bf16*)x, (const bf16*)w, rstd, (const bf16*)dres_in, (bf16*)dx_out, rows, d);
        else
            hipLaunchKernelGGL((rmsnorm_bwd_kernel<bf16, 256>), dim3(rows), dim3(256), 0, st, (const bf16*)dy,
                               (const bf16*)x, (const bf16*)w, rstd, (const bf16*)dres_in, (bf16*)dx_out, rows, d);
    }
    AV_LAUNCH_CHECK();
    return AV_OK;
}
