// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the AV->LLM hot path.
// Wavefront = 64 lanes everywhere; no other architecture is supported.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define AV_WAVE 64

// A/B switches (tools/*_bench.py, the tests that compare two forms of one computation): ONE table in api.hip, filled once per process
// from the environment (AVLLM_<NAME>) at first use and changed afterwards only through avllm_set_knob(name, value) -- never a getenv per
// layer or per launch.  Knobs that change what a production kernel computes or stores (GEMM_DBG) exist only in builds made with
// -DAVLLM_EXPERIMENT_KNOBS.
enum AvKnob { AV_KNOB_DECODE_FUSED, AV_KNOB_DEC_AL, AV_KNOB_LORA_UNBATCHED, AV_KNOB_F8_UNFUSED_QUANT, AV_KNOB_F8_FAST, AV_KNOB_ATTN_SHORT,
              AV_KNOB_NARROW_EPILOGUE, AV_KNOB_TN_CHUNK, AV_KNOB_GEMM_DBG, AV_KNOB_GEMM_GW, AV_KNOB_COUNT };
int av_knob(int id);

// ---- status / error string (thread local), SURVEY.md §8b "Errors" row
#define AV_OK 0
#define AV_ERR_ARG 1
#define AV_ERR_HIP 2
#define AV_ERR_WORKSPACE 3
#define AV_ERR_UNSUPPORTED 4

extern "C" const char* avllm_last_error(void);
int av_set_error(int code, const char* fmt, ...);

#define AV_CHECK_ARG(cond, ...) \
    do { if (!(cond)) return av_set_error(AV_ERR_ARG, __VA_ARGS__); } while (0)
#define AV_HIP(expr) \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) return av_set_error(AV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define AV_LAUNCH_CHECK() AV_HIP(hipGetLastError())
#define AV_TRY(expr) do { int rc_ = (expr); if (rc_ != AV_OK) return rc_; } while (0)

enum { AV_F32 = 0, AV_BF16 = 1 };
enum { AV_ACT_NONE = 0, AV_ACT_GELU = 1, AV_ACT_QUICK_GELU = 2, AV_ACT_SILU = 3 };

static inline size_t av_dtype_size(int dt) { return dt == AV_F32 ? 4 : 2; }

// ---- element helpers ---------------------------------------------------------------------
template <typename T> struct Vec;   // 16-byte vector of T

__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }

// load/store N consecutive elements (N*sizeof(T) must be 8 or 16 bytes and aligned)
template <typename T, int N> struct Pack { T v[N]; };

template <int N> __device__ __forceinline__ void load_f(const float* p, float (&o)[N]) {
    static_assert(N % 4 == 0, "");
#pragma unroll
    for (int i = 0; i < N; i += 4) { f32x4 t = *(const f32x4*)(p + i); o[i] = t[0]; o[i + 1] = t[1]; o[i + 2] = t[2]; o[i + 3] = t[3]; }
}
template <int N> __device__ __forceinline__ void load_f(const bf16* p, float (&o)[N]) {
    static_assert(N % 4 == 0, "");
    if constexpr (N % 8 == 0) {
#pragma unroll
        for (int i = 0; i < N; i += 8) { bf16x8 t = *(const bf16x8*)(p + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[i + j] = (float)t[j]; }
    } else {
#pragma unroll
        for (int i = 0; i < N; i += 4) { bf16x4 t = *(const bf16x4*)(p + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[i + j] = (float)t[j]; }
    }
}
template <int N> __device__ __forceinline__ void store_f(float* p, const float (&o)[N]) {
#pragma unroll
    for (int i = 0; i < N; i += 4) { f32x4 t = {o[i], o[i + 1], o[i + 2], o[i + 3]}; *(f32x4*)(p + i) = t; }
}
template <int N> __device__ __forceinline__ void store_f(bf16* p, const float (&o)[N]) {
    if constexpr (N % 8 == 0) {
#pragma unroll
        for (int i = 0; i < N; i += 8) { bf16x8 t;
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = (bf16)o[i + j];
            *(bf16x8*)(p + i) = t; }
    } else {
#pragma unroll
        for (int i = 0; i < N; i += 4) { bf16x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = (bf16)o[i + j];
            *(bf16x4*)(p + i) = t; }
    }
}

// ---- wave / block reductions ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// block reduction through LDS; `red` must hold >= blockDim.x/64 floats; all threads get the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case AV_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case AV_ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case AV_ACT_SILU: return x / (1.0f + __expf(-x));
        default: return x;
    }
}
// Same activations for the bf16 epilogues, where the result is rounded to 8 mantissa bits anyway: hardware exp2/rcp and the
// Abramowitz-Stegun 7.1.26 erf (|error| < 1.5e-7) instead of libm erff and IEEE division (3-4x fewer VALU instructions).
__device__ __forceinline__ float act_apply_fast(float x, int act) {
    switch (act) {
        case AV_ACT_GELU: {
            const float z = fabsf(x) * 0.70710678118654752f;
            const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
            const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
            const float erf_abs = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
            return 0.5f * x * (1.0f + copysignf(erf_abs, x));
        }
        case AV_ACT_QUICK_GELU: return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
        case AV_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
        default: return x;
    }
}

// Counter-based dropout mask: keep(seed, index) is a pure function, so forward and backward regenerate the same mask
// instead of storing it.  One 32-bit avalanche hash (lowbias32) serves TWO neighbouring elements (16 bits each), which keeps
// the mask cheap enough to be generated inside the rank-side GEMMs; the drop probability is therefore quantised to 1/65536
// (p = 0.05 -> 3276/65536) and the survivor scale uses that exact value.
__host__ __device__ __forceinline__ uint32_t av_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t av_drop_thr(float p) { return (uint32_t)(p * 65536.0f); }
__host__ __device__ __forceinline__ float av_drop_scale(float p) { return 1.0f / (1.0f - (float)av_drop_thr(p) * (1.0f / 65536.0f)); }
// hash word covering elements (2*pair, 2*pair+1)
// (the inner avalanche depends on the pair index only through its high word: written so that the common case -- high word 0, fewer than 2^33
// elements -- is av_hash32(seed), which is loop-invariant in every kernel and hoisted; one avalanche per pair instead of two.  Same value as
// av_hash32(lo ^ av_hash32(seed + hi * 0x9E3779B9)) for every index.)
__host__ __device__ __forceinline__ uint32_t av_pair_hash(uint32_t seed, unsigned long long pair) {
    const uint32_t hi = (uint32_t)(pair >> 32);
    uint32_t inner = av_hash32(seed);
    if (__builtin_expect(hi != 0u, 0)) inner = av_hash32(seed + hi * 0x9E3779B9U);
    return av_hash32((uint32_t)pair ^ inner);
}
__host__ __device__ __forceinline__ bool av_keep(uint32_t seed, unsigned long long idx, float p) {
    const uint32_t h = av_pair_hash(seed, idx >> 1);
    return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= av_drop_thr(p);
}
// The seed of a launch = (*seed_dev, when the caller keeps the step's base seed in device memory) + by-value offset.  A step whose
// kernels take every per-step scalar from device memory is the same launch sequence every time, i.e. it can be replayed from a hipGraph.
__device__ __forceinline__ uint32_t av_seed(const uint32_t* seed_dev, uint32_t off) { return (seed_dev ? *seed_dev : 0u) + off; }
// mask 8 consecutive elements starting at an EVEN index (4 hashes)
__device__ __forceinline__ void av_mask8(float (&v)[8], uint32_t seed, unsigned long long idx0, uint32_t thr, float sc) {
    const unsigned long long p0 = idx0 >> 1;
    if (__builtin_expect(((p0 + 3) >> 32) == 0, 1)) {      // one test per 8 elements: below 2^33 elements the inner avalanche is av_hash32(seed), a scalar the compiler hoists
        const uint32_t inner = av_hash32(seed);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t h = av_hash32(((uint32_t)p0 + q) ^ inner);
            v[2 * q] = (h & 0xffffu) >= thr ? v[2 * q] * sc : 0.f;
            v[2 * q + 1] = (h >> 16) >= thr ? v[2 * q + 1] * sc : 0.f;
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t h = av_pair_hash(seed, p0 + q);
        v[2 * q] = (h & 0xffffu) >= thr ? v[2 * q] * sc : 0.f;
        v[2 * q + 1] = (h >> 16) >= thr ? v[2 * q + 1] * sc : 0.f;
    }
}

static inline int av_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
