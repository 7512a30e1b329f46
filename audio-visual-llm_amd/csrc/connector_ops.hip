// Sequence operators of the non-default modality connectors (src/clip_whisper/models/modality_connector.py:111-380: `conv`, `attention`,
// `adaptive`), token-major [B, T, C] like every other activation of the path:
//   avllm_im2col_k3         nn.Conv1d(kernel_size 3, padding 1, stride 1 | 2) as im2col + avllm_gemm (the Whisper stem does the same)
//   avllm_groupnorm_tokens  nn.GroupNorm(G, C) of the [B, C, T] view: statistics over (T, C / G) per (item, group), optional GELU fused
// Byte movers plus one reduction; HBM-bound, far off the hot path (the default connector is one avllm_gemm).
#include "common.h"
#include "avllm_internal.h"

namespace {

// cols[(b, t'), kw * C + c] = x[b, stride * t' + kw - 1, c], zero outside [0, T)
template <typename T>
__global__ void im2col_k3_kernel(const T* __restrict__ x, T* __restrict__ cols, int B, int Tin, int Tout, int C, int stride) {
    const int c8 = C >> 3;
    const long total = (long)B * Tout * 3 * c8;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c8) * 8;
        const int kw = (int)((idx / c8) % 3);
        const long row = idx / ((long)c8 * 3);
        const int b = (int)(row / Tout), tp = (int)(row % Tout);
        const int t = stride * tp + kw - 1;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < Tin) load_f<8>(x + ((long)b * Tin + t) * C + c, v);
        store_f<8>(cols + row * 3 * C + (long)kw * C + c, v);
    }
}

// grid (G, B): one workgroup per (item, group); two passes over the [T, C/G] slab (second pass normalises, applies w, b and the activation)
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_tokens_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ bb,
                                                               T* __restrict__ y, int Tn, int C, int G, float eps, int act) {
    __shared__ float red[8];
    const int g = blockIdx.x, b = blockIdx.y, cg = C / G, c8 = cg >> 3;
    const T* xb = x + (long)b * Tn * C + (long)g * cg;
    T* yb = y + (long)b * Tn * C + (long)g * cg;
    const long n = (long)Tn * c8;
    float s = 0.f, ss = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        float v[8];
        load_f<8>(xb + (i / c8) * C + (i % c8) * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += v[j]; ss += v[j] * v[j]; }
    }
    const float cnt = (float)Tn * (float)cg;
    const float mean = block_sum(s, red) / cnt;
    const float var = fmaxf(block_sum(ss, red) / cnt - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    for (long i = threadIdx.x; i < n; i += 256) {
        const int c = (int)(i % c8) * 8;
        float v[8], wv[8], bv[8];
        load_f<8>(xb + (i / c8) * C + c, v);
        load_f<8>(w + (long)g * cg + c, wv);
        load_f<8>(bb + (long)g * cg + c, bv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act_apply((v[j] - mean) * rstd * wv[j] + bv[j], act);
        store_f<8>(yb + (i / c8) * C + c, v);
    }
}

inline int grid_1d(long total) {
    const long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int avllm_im2col_k3(const void* x, void* cols, int32_t B, int32_t T, int32_t C, int32_t stride, int32_t dtype, void* stream) {
    AV_CHECK_ARG(x && cols && B > 0 && T > 0 && C > 0 && C % 8 == 0 && (stride == 1 || stride == 2), "im2col_k3: bad args (C %% 8 == 0, stride 1 | 2)");
    const int Tout = (T - 1) / stride + 1;                          // (T + 2*1 - 3) / stride + 1
    const long total = (long)B * Tout * 3 * (C / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AV_F32) hipLaunchKernelGGL((im2col_k3_kernel<float>), dim3(grid_1d(total)), dim3(256), 0, st, (const float*)x, (float*)cols, B, T, Tout, C, stride);
    else hipLaunchKernelGGL((im2col_k3_kernel<bf16>), dim3(grid_1d(total)), dim3(256), 0, st, (const bf16*)x, (bf16*)cols, B, T, Tout, C, stride);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int avllm_groupnorm_tokens(const void* x, const void* w, const void* b, void* y, int32_t B, int32_t T, int32_t C, int32_t groups,
                                      float eps, int32_t act, int32_t dtype, void* stream) {
    AV_CHECK_ARG(x && w && b && y && B > 0 && B <= 65535 && T > 0 && groups > 0 && C % groups == 0 && (C / groups) % 8 == 0,
                 "groupnorm_tokens: bad args (C / groups must be a multiple of 8; C=%d groups=%d)", C, groups);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AV_F32) hipLaunchKernelGGL((groupnorm_tokens_kernel<float>), dim3(groups, B), dim3(256), 0, st, (const float*)x, (const float*)w, (const float*)b, (float*)y, T, C, groups, eps, act);
    else hipLaunchKernelGGL((groupnorm_tokens_kernel<bf16>), dim3(groups, B), dim3(256), 0, st, (const bf16*)x, (const bf16*)w, (const bf16*)b, (bf16*)y, T, C, groups, eps, act);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
