// LayerNorm / RMSNorm for gfx950.  HBM-bound: one pass over x (row kept in registers), 8/16-byte vector
// accesses, wave-shuffle reductions (one wave per row up to d=2048, one 256-thread block per row above).
// Reference arithmetic: nn.LayerNorm as used by HF whisper (:379-413) / clip (:362-384) encoder layers and
// LlamaRMSNorm (HF:models/llama/modeling_llama.py:62-67: fp32 variance, eps inside rsqrt).
#include "common.h"
#include "avllm_internal.h"

namespace {

constexpr int MAXV = 8;   // vectors per thread; VEC = 4 elements (fp32: 16 B, bf16: 8 B) or 8 (bf16: 16 B)

template <int NT> __device__ __forceinline__ float row_sum(float v, float* red) {
    if constexpr (NT == 64) return wave_sum(v);
    else return block_sum(v, red);
}

// MODE 0: layernorm (w,b)   MODE 1: rmsnorm fwd (w, writes rstd)
template <typename T, int NT, int MODE, int VEC>
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
    float v[MAXV][VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * NT + t) * VEC;
        if (c < d) {
            load_f<VEC>(xr + c, v[i]);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[i][j] = 0.f;
        }
    }
    float mean = 0.f;
    if (MODE == 0) mean = row_sum<NT>(s, red) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * NT + t) * VEC;
        if (c < d) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float u = v[i][j] - mean; q += u * u; }
        }
    }
    const float var = row_sum<NT>(q, red) / d;
    const float rstd = rsqrtf(var + eps);
    if (MODE == 1 && rstd_out && t == 0 && active) rstd_out[row] = rstd;
    if (!active) return;
    T* yr = y + row * d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * NT + t) * VEC;
        if (c < d) {
            float wv[VEC], o[VEC];
            load_f<VEC>(w + c, wv);
            if (MODE == 0) {
                float bv[VEC];
                load_f<VEC>(b + c, bv);
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = wv[j] * (v[i][j] * rstd);
            }
            store_f<VEC>(yr + c, o);
        }
    }
}

// LayerNorm for 768-wide bf16 rows (Whisper-small / CLIP ViT-B: 2/3 of all normalisation bytes of a step).  96 16-byte chunks do
// not fill a 64-lane wave evenly, so each lane takes one 16-byte vector of columns [0,512) and one 8-byte vector of [512,768):
// every lane busy, two loads instead of three, the wider one at the full 16 bytes.
__global__ __launch_bounds__(256) void layernorm768_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, const bf16* __restrict__ b,
                                                           bf16* __restrict__ y, long rows, float eps) {
    const int t = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16* xr = x + row * 768;
    float v8[8], v4[4];
    load_f<8>(xr + t * 8, v8);
    load_f<4>(xr + 512 + t * 4, v4);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v8[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v4[j];
    const float mean = wave_sum(s) * (1.0f / 768.0f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float u = v8[j] - mean; q += u * u; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float u = v4[j] - mean; q += u * u; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / 768.0f) + eps);
    float w8[8], b8[8], w4[4], b4[4], o8[8], o4[4];
    load_f<8>(w + t * 8, w8); load_f<8>(b + t * 8, b8);
    load_f<4>(w + 512 + t * 4, w4); load_f<4>(b + 512 + t * 4, b4);
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = (v8[j] - mean) * rstd * w8[j] + b8[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) o4[j] = (v4[j] - mean) * rstd * w4[j] + b4[j];
    bf16* yr = y + row * 768;
    store_f<8>(yr + t * 8, o8);
    store_f<4>(yr + 512 + t * 4, o4);
}

// dx = dres + rstd*(w*dy) - x*rstd^3*mean(w*dy*x)
template <typename T, int NT, int VEC>
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
    float xv[MAXV][VEC], gv[MAXV][VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * NT + t) * VEC;
        if (c < d) {
            float wv[VEC], dv[VEC];
            load_f<VEC>(x + r * d + c, xv[i]);
            load_f<VEC>(dy + r * d + c, dv);
            load_f<VEC>(w + c, wv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { gv[i][j] = wv[j] * dv[j]; s += gv[i][j] * xv[i][j]; }
        }
    }
    const float dot = row_sum<NT>(s, red);
    if (!active) return;
    const float rs = rstd[row];
    const float coef = dot * rs * rs * rs / d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * NT + t) * VEC;
        if (c < d) {
            float o[VEC], dr[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) dr[j] = 0.f;
            if (dres) load_f<VEC>(dres + row * d + c, dr);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = dr[j] + rs * gv[i][j] - xv[i][j] * coef;
            store_f<VEC>(dx + row * d + c, o);
        }
    }
}

template <typename T, int MODE, int VEC>
int launch_fwd_v(const void* x, const void* w, const void* b, void* y, float* rstd, long rows, int d, float eps, hipStream_t st) {
    if (d <= 64 * VEC * MAXV) {
        hipLaunchKernelGGL((norm_fwd_kernel<T, 64, MODE, VEC>), dim3(av_cdiv(rows, 4)), dim3(256), 0, st, (const T*)x, (const T*)w,
                           (const T*)b, (T*)y, rstd, rows, d, eps);
    } else {
        hipLaunchKernelGGL((norm_fwd_kernel<T, 256, MODE, VEC>), dim3(rows), dim3(256), 0, st, (const T*)x, (const T*)w,
                           (const T*)b, (T*)y, rstd, rows, d, eps);
    }
    AV_LAUNCH_CHECK();
    return AV_OK;
}
// bf16 rows that fill whole waves with 16-byte vectors (d % 512 == 0, e.g. Llama's 4096: one wave per row) use them; shorter rows
// such as 768 keep 8-byte vectors with every lane busy (measured: 16-byte vectors with 32 idle lanes are 23 % slower there)
template <typename T, int MODE>
int launch_fwd(const void* x, const void* w, const void* b, void* y, float* rstd, long rows, int d, float eps, hipStream_t st) {
    if (sizeof(T) == 2 && d % 512 == 0) return launch_fwd_v<T, MODE, 8>(x, w, b, y, rstd, rows, d, eps, st);
    return launch_fwd_v<T, MODE, 4>(x, w, b, y, rstd, rows, d, eps, st);
}

template <typename T, int VEC>
void launch_rms_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres_in, void* dx_out, long rows, int d, hipStream_t st) {
    if (d <= 64 * VEC * MAXV)
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<T, 64, VEC>), dim3(av_cdiv(rows, 4)), dim3(256), 0, st, (const T*)dy, (const T*)x, (const T*)w, rstd,
                           (const T*)dres_in, (T*)dx_out, rows, d);
    else
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<T, 256, VEC>), dim3(rows), dim3(256), 0, st, (const T*)dy, (const T*)x, (const T*)w, rstd,
                           (const T*)dres_in, (T*)dx_out, rows, d);
}

}  // namespace

int av_layernorm(const void* x, const void* w, const void* b, void* y, long rows, int d, float eps, int dtype, hipStream_t st) {
    AV_CHECK_ARG(x && w && b && y && rows > 0, "layernorm: null/empty");
    AV_CHECK_ARG(d % 4 == 0 && d <= 256 * 4 * MAXV, "layernorm: d=%d unsupported", d);
    if (dtype == AV_BF16 && d == 768) {
        hipLaunchKernelGGL(layernorm768_kernel, dim3(av_cdiv(rows, 4)), dim3(256), 0, st, (const bf16*)x, (const bf16*)w, (const bf16*)b, (bf16*)y, rows, eps);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
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
    if (dtype == AV_F32) launch_rms_bwd<float, 4>(dy, x, w, rstd, dres_in, dx_out, rows, d, st);
    else if (d % 512 == 0) launch_rms_bwd<bf16, 8>(dy, x, w, rstd, dres_in, dx_out, rows, d, st);
    else launch_rms_bwd<bf16, 4>(dy, x, w, rstd, dres_in, dx_out, rows, d, st);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
