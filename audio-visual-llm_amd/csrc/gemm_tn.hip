// LoRA gradient reductions on MFMA (bf16):  Z[n,j] = alpha * sum_m Big[m,n] * Small[m,j],  n < NB (model width),
// j < 16 (rank).  dB = dY^T.(s x A^T) uses Z as is; dA = (s dY B)^T.x stores Z transposed (autograd of peft lora.Linear,
// reference wrap clip_whisper_model.py:961-1005).  Both operands are row-major over the REDUCTION index m, i.e. k-strided
// for an MFMA, so both go through LDS and come back as fragments with ds_read_b64_tr_b16 (hardware transpose).
// Grid: (NB/128, M-chunks); fp32 atomics accumulate the chunk partials (<= 32 adders per element).
#include "common.h"
#include "avllm_internal.h"
#include <cstdlib>

namespace {

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;
typedef __attribute__((ext_vector_type(8))) short short8v;

// 16x16x32 operand whose k index runs over LDS tile rows [row0, row0+32) and whose 16 rows/cols are tile columns col0..col0+15
__device__ __forceinline__ bf16x8 tr_frag16(const char* img, int stride, int row0, int col0, int lane) {
    const int g = lane >> 4, i16 = lane & 15;
    const char* a0 = img + (row0 + 8 * g + (i16 >> 2)) * stride + (col0 + 4 * (i16 & 3)) * 2;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0 + 4 * stride));
    const short8v both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

constexpr int TN_N = 128, TN_M = 64;
constexpr int BIG_STRIDE = TN_N * 2 + 64;     // 320 B: the 4 rows of a transposed read land on distinct bank quarters
constexpr int SM_STRIDE = 48;                 // 16 bf16 + pad

template <bool TRANS_OUT>
__global__ __launch_bounds__(256) void gemm_tn_mfma_kernel(const bf16* __restrict__ Big, long ldb, const bf16* __restrict__ Small,
                                                           long lds_, int R, int M, int mchunk, float* __restrict__ out, long ldo,
                                                           float alpha, int NB, uint32_t drop_seed_off, float drop_p, const uint32_t* seed_dev) {
    __shared__ __attribute__((aligned(16))) char big_s[TN_M * BIG_STRIDE];
    __shared__ __attribute__((aligned(16))) char small_s[TN_M * SM_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n0 = blockIdx.x * TN_N;
    const uint32_t drop_seed = drop_p > 0.f ? av_seed(seed_dev, drop_seed_off) : 0u;
    const int m_begin = blockIdx.y * mchunk, m_end = min(M, m_begin + mchunk);
    f32x4 acc[2];
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // software pipeline: the next 64-row slab travels global -> registers while the MFMAs of the current one run
    constexpr int BIG_PER_THREAD = TN_M * (TN_N / 8) / 256;       // 4 16-byte chunks of the wide operand per thread and slab
    u32x4 nb[BIG_PER_THREAD], ns = {0u, 0u, 0u, 0u};
    auto fetch = [&](int mb) {
#pragma unroll
        for (int i = 0; i < BIG_PER_THREAD; ++i) {
            const int c = tid + i * 256, row = c >> 4, ch = c & 15;
            nb[i] = (u32x4){0u, 0u, 0u, 0u};
            if (mb + row < m_end) nb[i] = *(const u32x4*)(Big + (long)(mb + row) * ldb + n0 + ch * 8);
        }
        if (tid < TN_M * 2) {
            const int row = tid >> 1, ch = tid & 1;
            ns = (u32x4){0u, 0u, 0u, 0u};
            if (mb + row < m_end) ns = *(const u32x4*)(Small + (long)(mb + row) * lds_ + ch * 8);
        }
    };
    if (m_begin < m_end) fetch(m_begin);
    for (int mb = m_begin; mb < m_end; mb += TN_M) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BIG_PER_THREAD; ++i) {
            const int c = tid + i * 256, row = c >> 4, ch = c & 15;
            u32x4 v = nb[i];
            if (drop_p > 0.f) {          // Big = dropout(x): regenerate the forward's mask (index row*NB + col) instead of reading a copy
                const bf16x8 xb = __builtin_bit_cast(bf16x8, v);
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (float)xb[j];
                av_mask8(f, drop_seed, (unsigned long long)(mb + row) * NB + n0 + ch * 8, av_drop_thr(drop_p), av_drop_scale(drop_p));
                bf16x8 yb;
#pragma unroll
                for (int j = 0; j < 8; ++j) yb[j] = (bf16)f[j];
                v = __builtin_bit_cast(u32x4, yb);
            }
            *(u32x4*)(big_s + row * BIG_STRIDE + ch * 16) = v;
        }
        if (tid < TN_M * 2) *(u32x4*)(small_s + (tid >> 1) * SM_STRIDE + (tid & 1) * 16) = ns;
        __syncthreads();
        if (mb + TN_M < m_end) fetch(mb + TN_M);
#pragma unroll
        for (int ks = 0; ks < TN_M / 32; ++ks) {
            const bf16x8 sa = tr_frag16(small_s, SM_STRIDE, 32 * ks, 0, lane);                   // A: rows j, k = m
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 bb = tr_frag16(big_s, BIG_STRIDE, 32 * ks, w * 32 + t * 16, lane);  // B: k = m, cols n
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, bb, acc[t], 0, 0, 0);        // D[j][n]
            }
        }
    }
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + w * 32 + t * 16 + fr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = fq * 4 + i;
            if (j < R) {
                float* p = TRANS_OUT ? out + (long)j * ldo + n : out + (long)n * ldo + j;
                atomicAdd(p, alpha * acc[t][i]);
            }
        }
    }
}

}  // namespace

// Big [M,NB] (NB % 128 == 0), Small [M,>=16 cols, R valid]; out [NB,R] (TRANS_OUT=0) or [R,NB] (TRANS_OUT=1)
int av_gemm_tn_mfma(const void* Big, long ldb, int NB, const void* Small, long lds_, int R, int M, float* out, long ldo, float alpha,
                    int trans_out, hipStream_t st, uint32_t drop_seed, float drop_p, const uint32_t* seed_dev) {
    // token chunks: enough workgroups to fill the chip (NB/128 x zs >= 256) but as few atomic adders per element as that allows
    const int chunk_env = av_knob(AV_KNOB_TN_CHUNK);
    // [NB,R] output = 64-byte rows scattered across lanes (slow atomics): fewer, longer chunks (measured 13.8 vs 18.3 us at 512 vs 256)
    const int want = chunk_env > 0 ? chunk_env : (trans_out ? 256 : 512);
    int zs = av_cdiv(M, want);
    zs = zs > 32 ? 32 : zs;
    int mchunk = av_cdiv(M, zs);
    mchunk = (mchunk + TN_M - 1) / TN_M * TN_M;
    zs = av_cdiv(M, mchunk);
    const dim3 grid(NB / TN_N, zs);
    if (trans_out) hipLaunchKernelGGL((gemm_tn_mfma_kernel<true>), grid, dim3(256), 0, st, (const bf16*)Big, ldb, (const bf16*)Small, lds_, R, M, mchunk, out, ldo, alpha, NB, drop_seed, drop_p, seed_dev);
    else hipLaunchKernelGGL((gemm_tn_mfma_kernel<false>), grid, dim3(256), 0, st, (const bf16*)Big, ldb, (const bf16*)Small, lds_, R, M, mchunk, out, ldo, alpha, NB, drop_seed, drop_p, seed_dev);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
