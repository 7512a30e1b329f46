// The q / k / v adapters of one decoder layer in ONE launch each way (bf16).  peft wraps q_proj, k_proj and v_proj separately
// (reference wrap clip_whisper_model.py:961-1005), but the three adapters read the SAME normed input and their output gradients are
// column slices of ONE dqkv buffer, so per layer
//     forward   t_j  = s * dropout_j(x) A_j^T                 3 launches reading x three times     -> 1 (x read once, three masks)
//     backward  dt_j = s * dy_j B_j                           3 launches                            -> 1 (grid.y = adapter)
//               dB_j = dy_j^T t_j                             3 launches of 256 workgroups          -> 1 of 768
//               dA_j = dt_j^T dropout_j(x)                    3 launches reading x three times     -> 1 (x read once)
// Same arithmetic as gemm_skinny64_kernel / gemm_tn_mfma_kernel (gemm.hip, gemm_tn.hip), same mask function (common.h av_mask8).
#include "common.h"
#include "avllm_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------------------ rank side: C_j[M,64] = alpha A_j[M,K_j] B_j[R<=16,K_j]^T
constexpr int RK_WAVES = 8;

struct Rank3Args {
    const bf16* A[3]; long lda[3]; int K[3];
    const bf16* B[3]; long ldb[3];
    bf16* C[3]; long ldc[3];
    uint32_t seed[3];
    int nj, M;
    float alpha, p;
    const uint32_t* seed_dev;
};

__device__ __forceinline__ bf16x8 rk_drop(bf16x8 x, uint32_t seed, unsigned long long idx0, uint32_t thr, float sc) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
    av_mask8(v, seed, idx0, thr, sc);
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16)v[j];
    return r;
}

// SHARED: every adapter reads A[0] (the three forward products of one input; DROP applies adapter j's mask to the shared fragment).
// !SHARED: blockIdx.y picks the adapter (its own A, K, B).  A workgroup owns 16 rows, its 8 waves split K, partials meet in LDS;
// the 48 padding columns of every 64-wide output are written as zeros (they are the K2 segment of the projection GEMM).
template <bool SHARED, bool DROP, int NJ>
__global__ __launch_bounds__(RK_WAVES * 64) void lora_rank3_kernel(Rank3Args a) {
    __shared__ float part[NJ][RK_WAVES][16][17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int j0 = SHARED ? 0 : blockIdx.y;
    int ar = m0 + fr; ar = ar < a.M ? ar : a.M - 1;
    const int K = a.K[j0], kw = K / RK_WAVES;                      // multiple of 32
    const bf16* ap = a.A[j0] + (long)ar * a.lda[j0] + (long)w * kw + fq * 8;
    const bf16* bp[NJ];
    uint32_t seed[NJ];
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        bp[j] = a.B[j0 + j] + (long)fr * a.ldb[j0 + j] + (long)w * kw + fq * 8;
        seed[j] = DROP ? av_seed(a.seed_dev, a.seed[j0 + j]) : 0u;
        acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const uint32_t thr = av_drop_thr(a.p);
    const float dsc = av_drop_scale(a.p);
    const unsigned long long idx_base = (unsigned long long)ar * K + (unsigned long long)w * kw + fq * 8;     // mask index = row * K + col
    constexpr int UNR = 4;
    int k = 0;
    for (; k + 32 * UNR <= kw; k += 32 * UNR) {
        bf16x8 xa[UNR], wb[UNR][NJ];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            xa[u] = *(const bf16x8*)(ap + k + 32 * u);
#pragma unroll
            for (int j = 0; j < NJ; ++j) wb[u][j] = *(const bf16x8*)(bp[j] + k + 32 * u);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bf16x8 x = DROP ? rk_drop(xa[u], seed[j], idx_base + k + 32 * u, thr, dsc) : xa[u];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[u][j], x, acc[j], 0, 0, 0);       // D[n][m]
            }
    }
    for (; k < kw; k += 32) {
        const bf16x8 xa = *(const bf16x8*)(ap + k);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bf16x8 wb = *(const bf16x8*)(bp[j] + k);
            const bf16x8 x = DROP ? rk_drop(xa, seed[j], idx_base + k, thr, dsc) : xa;
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, x, acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[j][w][fr][fq * 4 + i] = acc[j][i];          // [m][n]
    __syncthreads();
    for (int e = threadIdx.x; e < NJ * 16 * 64; e += RK_WAVES * 64) {
        const int j = e / (16 * 64), r = (e >> 6) & 15, c = e & 63;
        if (m0 + r >= a.M) continue;
        float s = 0.f;
        if (c < 16) {
#pragma unroll
            for (int x = 0; x < RK_WAVES; ++x) s += part[j][x][r][c];
            s *= a.alpha;
        }
        a.C[j0 + j][(long)(m0 + r) * a.ldc[j0 + j] + c] = (bf16)s;
    }
}

// ------------------------------------------------------------------------------------------------------------ reductions over tokens
typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;
typedef __attribute__((ext_vector_type(8))) short short8v;

__device__ __forceinline__ bf16x8 tn_frag16(const char* img, int stride, int row0, int col0, int lane) {
    const int g = lane >> 4, i16 = lane & 15;
    const char* a0 = img + (row0 + 8 * g + (i16 >> 2)) * stride + (col0 + 4 * (i16 & 3)) * 2;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0 + 4 * stride));
    const short8v both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

constexpr int TM_N = 128, TM_M = 64, TM_BIG = TM_N * 2 + 64, TM_SM = 48;      // as gemm_tn.hip: 64-row slabs, 320-byte / 48-byte LDS rows

struct TnMultiArgs {
    const bf16* Big; long ldb; int NB;            // wide operand [M, NB]
    const bf16* Small[3]; long lds[3];            // [M, >= 16] each
    float* out[3]; long ldo[3];
    int col0[3], ncol[3];                         // !SHARED: adapter j owns Big's columns [col0, col0 + ncol)
    uint32_t seed[3];
    int nj, R, M, mchunk;
    float alpha, p;
    const uint32_t* seed_dev;
};

// SHARED (dA: out_j[R, NB] += Small_j^T . mask_j(Big)): one slab of Big is loaded once and staged as NJ masked images.
// !SHARED (dB: out_j[ncol_j, R] += Big[:, cols_j]^T . Small_j): the workgroup's 128 columns lie inside one adapter's range.
template <bool SHARED, int NJ>
__global__ __launch_bounds__(256) void gemm_tn_multi_kernel(TnMultiArgs a) {
    extern __shared__ __attribute__((aligned(16))) char tm_smem[];
    char* big_s = tm_smem;                                        // NJ images of TM_M x TM_BIG
    char* small_s = tm_smem + NJ * TM_M * TM_BIG;                 // NJ images of TM_M x TM_SM
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n0 = blockIdx.x * TM_N;
    int j0 = 0;
    if (!SHARED) j0 = (a.nj > 1 && n0 >= a.col0[1]) + (a.nj > 2 && n0 >= a.col0[2]);
    const int m_begin = blockIdx.y * a.mchunk, m_end = min(a.M, m_begin + a.mchunk);
    const bool drop = SHARED && a.p > 0.f;
    uint32_t seed[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) seed[j] = drop ? av_seed(a.seed_dev, a.seed[j]) : 0u;
    f32x4 acc[NJ][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    constexpr int BPT = TM_M * (TM_N / 8) / 256;                  // 4 chunks of the wide operand per thread and slab
    u32x4 nb[BPT], ns[NJ];
    auto fetch = [&](int mb) {
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int c = tid + i * 256, row = c >> 4, ch = c & 15;
            nb[i] = (u32x4){0u, 0u, 0u, 0u};
            if (mb + row < m_end) nb[i] = *(const u32x4*)(a.Big + (long)(mb + row) * a.ldb + n0 + ch * 8);
        }
        if (tid < TM_M * 2) {
            const int row = tid >> 1, ch = tid & 1;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                ns[j] = (u32x4){0u, 0u, 0u, 0u};
                if (mb + row < m_end) ns[j] = *(const u32x4*)(a.Small[j0 + j] + (long)(mb + row) * a.lds[j0 + j] + ch * 8);
            }
        }
    };
    if (m_begin < m_end) fetch(m_begin);
    const uint32_t thr = av_drop_thr(a.p);
    const float dsc = av_drop_scale(a.p);
    for (int mb = m_begin; mb < m_end; mb += TM_M) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int c = tid + i * 256, row = c >> 4, ch = c & 15;
            if (drop) {
                const bf16x8 xb = __builtin_bit_cast(bf16x8, nb[i]);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float f[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) f[q] = (float)xb[q];
                    av_mask8(f, seed[j], (unsigned long long)(mb + row) * a.NB + n0 + ch * 8, thr, dsc);
                    bf16x8 yb;
#pragma unroll
                    for (int q = 0; q < 8; ++q) yb[q] = (bf16)f[q];
                    *(u32x4*)(big_s + j * TM_M * TM_BIG + row * TM_BIG + ch * 16) = __builtin_bit_cast(u32x4, yb);
                }
            } else {
                *(u32x4*)(big_s + row * TM_BIG + ch * 16) = nb[i];           // one image serves every adapter
            }
        }
        if (tid < TM_M * 2) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) *(u32x4*)(small_s + j * TM_M * TM_SM + (tid >> 1) * TM_SM + (tid & 1) * 16) = ns[j];
        }
        __syncthreads();
        if (mb + TM_M < m_end) fetch(mb + TM_M);
#pragma unroll
        for (int ks = 0; ks < TM_M / 32; ++ks) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bf16x8 sa = tn_frag16(small_s + j * TM_M * TM_SM, TM_SM, 32 * ks, 0, lane);                       // A: rows r, k = m
                const char* img = big_s + (drop ? j * TM_M * TM_BIG : 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 bb = tn_frag16(img, TM_BIG, 32 * ks, w * 32 + t * 16, lane);                           // B: k = m, cols n
                    acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, bb, acc[j][t], 0, 0, 0);                    // D[r][n]
                }
            }
        }
    }
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = n0 + w * 32 + t * 16 + fr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = fq * 4 + i;
                if (r < a.R) {
                    float* p = SHARED ? a.out[j] + (long)r * a.ldo[j] + n : a.out[j0] + (long)(n - a.col0[j0]) * a.ldo[j0] + r;
                    atomicAdd(p, a.alpha * acc[j][t][i]);
                }
            }
        }
}

int tn_chunks(int M, int want, int& mchunk) {
    int zs = av_cdiv(M, want);
    zs = zs > 32 ? 32 : zs;
    mchunk = av_cdiv(M, zs);
    mchunk = (mchunk + TM_M - 1) / TM_M * TM_M;
    return av_cdiv(M, mchunk);
}

}  // namespace

bool av_lora_batch_supported(int dtype, int R, int nj) { return dtype == AV_BF16 && R >= 1 && R <= 16 && nj >= 1 && nj <= 3; }

// forward (shared != 0): C_j = alpha * dropout_j(A_0) B_j^T; backward (shared == 0): C_j = alpha * A_j B_j^T
int av_lora_rank3(const void* const* A, const long* lda, const int* K, const void* const* B, const long* ldb, void* const* C, const long* ldc,
                  const uint32_t* seeds, int nj, int M, int R, float alpha, float p, const uint32_t* seed_dev, int shared, int dtype, hipStream_t st) {
    AV_CHECK_ARG(A && lda && K && B && ldb && C && ldc && M > 0, "lora_rank3: null/empty");
    AV_CHECK_ARG(av_lora_batch_supported(dtype, R, nj), "lora_rank3: bf16, rank <= 16 (r=%d), 1..3 adapters (nj=%d)", R, nj);
    AV_CHECK_ARG(p >= 0.f && p < 1.f && (p == 0.f || shared), "lora_rank3: dropout (p=%f) only on the shared-input form", p);
    Rank3Args a = {};
    for (int j = 0; j < nj; ++j) {
        const int ja = shared ? 0 : j;
        AV_CHECK_ARG(A[ja] && B[j] && C[j], "lora_rank3: null operand %d", j);
        AV_CHECK_ARG(K[ja] > 0 && K[ja] % (32 * RK_WAVES) == 0 && lda[ja] % 8 == 0 && ldb[j] % 8 == 0 && ldc[j] >= 64,
                     "lora_rank3: K %% 256 == 0 (K=%d), 16-byte rows, 64-wide outputs", K[ja]);
        AV_CHECK_ARG(p == 0.f || lda[ja] == K[ja], "lora_rank3: the fused mask needs the full contiguous [M,K] activation as A");
        a.A[j] = (const bf16*)A[ja]; a.lda[j] = lda[ja]; a.K[j] = K[ja]; a.B[j] = (const bf16*)B[j]; a.ldb[j] = ldb[j];
        a.C[j] = (bf16*)C[j]; a.ldc[j] = ldc[j]; a.seed[j] = seeds ? seeds[j] : 0u;
    }
    a.nj = nj; a.M = M; a.alpha = alpha; a.p = p; a.seed_dev = seed_dev;
    const int gx = av_cdiv(M, 16);
    const dim3 blk(RK_WAVES * 64);
#define RK_LAUNCH(SH, DR, NJV, GY) hipLaunchKernelGGL((lora_rank3_kernel<SH, DR, NJV>), dim3(gx, GY), blk, 0, st, a)
    if (shared) {
        if (p > 0.f) { if (nj == 3) RK_LAUNCH(true, true, 3, 1); else if (nj == 2) RK_LAUNCH(true, true, 2, 1); else RK_LAUNCH(true, true, 1, 1); }
        else { if (nj == 3) RK_LAUNCH(true, false, 3, 1); else if (nj == 2) RK_LAUNCH(true, false, 2, 1); else RK_LAUNCH(true, false, 1, 1); }
    } else RK_LAUNCH(false, false, 1, nj);
#undef RK_LAUNCH
    AV_LAUNCH_CHECK();
    return AV_OK;
}

// shared != 0: out_j[R, NB] += alpha * Small_j^T . dropout_j(Big)   (Big [M, NB], NB % 128 == 0)
// shared == 0: out_j[ncol_j, R] += alpha * Big[:, col0_j : col0_j + ncol_j]^T . Small_j   (col0_j, ncol_j multiples of 128, ranges ascending and adjacent)
int av_gemm_tn_multi(const void* Big, long ldb, int NB, const void* const* Small, const long* lds, float* const* out, const long* ldo,
                     const int* col0, const int* ncol, const uint32_t* seeds, int nj, int R, int M, float alpha, float p,
                     const uint32_t* seed_dev, int shared, int dtype, hipStream_t st) {
    AV_CHECK_ARG(Big && Small && lds && out && ldo && M > 0 && NB > 0 && NB % TM_N == 0 && ldb % 8 == 0, "gemm_tn_multi: bad args (NB=%d)", NB);
    AV_CHECK_ARG(av_lora_batch_supported(dtype, R, nj), "gemm_tn_multi: bf16, rank <= 16 (r=%d), 1..3 adapters (nj=%d)", R, nj);
    AV_CHECK_ARG(p >= 0.f && p < 1.f && (p == 0.f || shared), "gemm_tn_multi: dropout (p=%f) only on the shared form", p);
    TnMultiArgs a = {};
    a.Big = (const bf16*)Big; a.ldb = ldb; a.NB = NB;
    int next = 0;
    for (int j = 0; j < nj; ++j) {
        AV_CHECK_ARG(Small[j] && out[j] && lds[j] % 8 == 0 && lds[j] >= 16, "gemm_tn_multi: operand %d", j);
        a.Small[j] = (const bf16*)Small[j]; a.lds[j] = lds[j]; a.out[j] = out[j]; a.ldo[j] = ldo[j]; a.seed[j] = seeds ? seeds[j] : 0u;
        if (!shared) {
            AV_CHECK_ARG(col0 && ncol && col0[j] == next && ncol[j] > 0 && ncol[j] % TM_N == 0, "gemm_tn_multi: column ranges must tile [0, NB) in 128-column units");
            a.col0[j] = col0[j]; a.ncol[j] = ncol[j]; next += ncol[j];
        }
    }
    AV_CHECK_ARG(shared || next == NB, "gemm_tn_multi: column ranges cover %d of %d columns", next, NB);
    a.nj = nj; a.R = R; a.M = M; a.alpha = alpha; a.p = p; a.seed_dev = seed_dev;
    const int zs = tn_chunks(M, shared ? 256 : 512, a.mchunk);          // as gemm_tn.hip: [ncol, R] outputs take fewer, longer chunks
    const dim3 grid(NB / TM_N, zs);
    const int njk = shared ? nj : 1;
    const size_t lds_bytes = (size_t)njk * TM_M * (TM_BIG + TM_SM);
    static bool attr[64][3] = {};
    int dev = 0;
    AV_HIP(hipGetDevice(&dev));
#define TM_LAUNCH(SH, NJV) do { \
        if (!attr[dev & 63][NJV - 1] && SH) { AV_HIP(hipFuncSetAttribute((const void*)gemm_tn_multi_kernel<SH, NJV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); attr[dev & 63][NJV - 1] = true; } \
        hipLaunchKernelGGL((gemm_tn_multi_kernel<SH, NJV>), grid, dim3(256), lds_bytes, st, a); } while (0)
    if (shared) { if (nj == 3) TM_LAUNCH(true, 3); else if (nj == 2) TM_LAUNCH(true, 2); else TM_LAUNCH(true, 1); }
    else TM_LAUNCH(false, 1);
#undef TM_LAUNCH
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int avllm_lora_rank3(const void* const* A, const int64_t* lda, const int32_t* K, const void* const* B, const int64_t* ldb, void* const* C,
                                const int64_t* ldc, const uint32_t* seeds, int32_t nj, int32_t M, int32_t R, float alpha, float p,
                                const uint32_t* seed_dev, int32_t shared, int32_t dtype, void* stream) {
    AV_CHECK_ARG(nj >= 1 && nj <= 3 && lda && K && ldb && ldc, "lora_rank3: nj=%d", nj);
    long la[3], lb[3], lc[3]; int kk[3];
    for (int j = 0; j < nj; ++j) { la[j] = (long)lda[j]; lb[j] = (long)ldb[j]; lc[j] = (long)ldc[j]; kk[j] = K[j]; }
    return av_lora_rank3(A, la, kk, B, lb, C, lc, seeds, nj, M, R, alpha, p, seed_dev, shared, dtype, (hipStream_t)stream);
}

extern "C" int avllm_gemm_tn_multi(const void* Big, int64_t ldb, int32_t NB, const void* const* Small, const int64_t* lds, float* const* out,
                                   const int64_t* ldo, const int32_t* col0, const int32_t* ncol, const uint32_t* seeds, int32_t nj, int32_t R,
                                   int32_t M, float alpha, float p, const uint32_t* seed_dev, int32_t shared, int32_t dtype, void* stream) {
    AV_CHECK_ARG(nj >= 1 && nj <= 3 && lds && ldo, "gemm_tn_multi: nj=%d", nj);
    long ls[3], lo[3]; int c0[3] = {0, 0, 0}, nc[3] = {0, 0, 0};
    for (int j = 0; j < nj; ++j) { ls[j] = (long)lds[j]; lo[j] = (long)ldo[j]; if (col0) c0[j] = col0[j]; if (ncol) nc[j] = ncol[j]; }
    return av_gemm_tn_multi(Big, ldb, NB, Small, ls, out, lo, col0 ? c0 : nullptr, ncol ? nc : nullptr, seeds, nj, R, M, alpha, p, seed_dev, shared,
                            dtype, (hipStream_t)stream);
}
