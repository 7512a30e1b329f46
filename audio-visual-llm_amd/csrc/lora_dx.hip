// Adapter input gradient under LoRA dropout, all adapters of one input in ONE pass (bf16):
//
//     out[M,N] = R[M,N] + sum_{j < nj} mask_j o ( T_j[M,r] . A_j[r,N] ) / (1-p)
//
// peft's lora.Linear computes lora_B(lora_A(dropout(x))) (reference wrap clip_whisper_model.py:961-1005), so the gradient that
// reaches x through adapter j is mask_j o (dt_j . A_j): the mask sits on the OUTPUT of a rank-r product and cannot ride in the big
// dX GEMM's accumulator.  Round 1 ran one K=64 GEMM with a masked epilogue per adapter (3 read-modify-write passes over dX for
// q/k/v: 3 x 64 MB at M = N = 4096, memory-bound); here the nj rank-r products of a 16x16 output tile are nj single MFMAs whose
// results are masked in the accumulator layout, summed, and leave through a wave-private LDS transpose as 16-byte row chunks:
// dX is read once and written once, and rounded once.
//
// T_j: [M, >=32 cols] bf16, columns >= r are zero (the padded rank-side output of engine.hip); A_j is given as the padded
// transposed image AT_j [N, >=32 cols] (row n = column n of A_j, zeros past r), as avllm_lora_pack lays it out.
#include "common.h"
#include "avllm_internal.h"

namespace {

constexpr int DX_ROWS = 32, DX_COLS = 128;      // per wave; a workgroup = 4 waves stacked in M

struct LoraDxArgs {
    const bf16* T[3]; const bf16* AT[3];
    long ldt[3], ldat[3];
    uint32_t seed[3];
    const bf16* R; bf16* out;
    long ldr, ldo;
    int M, N, nj;
    float p;
    const uint32_t* seed_dev;
};

// mask 4 consecutive elements starting at an index that is a multiple of 4 (2 pair hashes)
__device__ __forceinline__ void mask4_add(f32x4& acc, const f32x4 v, uint32_t seed, unsigned long long idx0, uint32_t thr, float sc) {
    const uint32_t h0 = av_pair_hash(seed, idx0 >> 1), h1 = av_pair_hash(seed, (idx0 >> 1) + 1);
    acc[0] += (h0 & 0xffffu) >= thr ? v[0] * sc : 0.f;
    acc[1] += (h0 >> 16) >= thr ? v[1] * sc : 0.f;
    acc[2] += (h1 & 0xffffu) >= thr ? v[2] * sc : 0.f;
    acc[3] += (h1 >> 16) >= thr ? v[3] * sc : 0.f;
}

template <int NJ>
__global__ __launch_bounds__(256) void lora_dx_masked_kernel(LoraDxArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[4 * DX_ROWS * DX_COLS];        // 64 KiB: one 32x128 fp32 slice per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = (blockIdx.y * 4 + wave) * DX_ROWS, n0 = blockIdx.x * DX_COLS;
    if (m0 >= a.M) return;                                                            // wave-uniform; no workgroup barrier below
    float* cw = lds + wave * DX_ROWS * DX_COLS;
    const uint32_t thr = av_drop_thr(a.p);
    const float sc = av_drop_scale(a.p);
    uint32_t seed[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) seed[j] = av_seed(a.seed_dev, a.seed[j]);
    // activation-side fragments: T_j[m][8 fq .. +7] (k < 32; columns past the rank are zeros)
    bf16x8 xa[2][NJ];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        int m = m0 + rb * 16 + fr;
        m = m < a.M ? m : a.M - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) xa[rb][j] = *(const bf16x8*)(a.T[j] + (long)m * a.ldt[j] + fq * 8);
    }
#pragma unroll 2
    for (int ct = 0; ct < DX_COLS / 16; ++ct) {
        bf16x8 wb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) wb[j] = *(const bf16x8*)(a.AT[j] + (long)(n0 + ct * 16 + fr) * a.ldat[j] + fq * 8);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int row = rb * 16 + fr;
            // D[n][m]: this lane holds output row m = m0 + row, columns n0 + 16 ct + 4 fq .. +3
            const unsigned long long idx0 = (unsigned long long)(m0 + row) * a.N + n0 + ct * 16 + 4 * fq;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 pj = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[rb][j], z, 0, 0, 0);
                mask4_add(acc, pj, seed[j], idx0, thr, sc);
            }
            *(f32x4*)(cw + row * DX_COLS + (((ct * 4 + fq) ^ (row & 7)) << 2)) = acc;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same-wave LDS traffic is ordered; this pins the compiler's ordering
    // out: 32 rows x 16 chunks of 8 columns, 8 chunks per lane; one wave instruction = 4 rows x 256 contiguous bytes
#pragma unroll
    for (int i = 0; i < DX_ROWS * (DX_COLS / 8) / 64; ++i) {
        const int c = lane + 64 * i, row = c >> 4, ch = c & 15;
        const int m = m0 + row;
        const f32x4 lo = *(const f32x4*)(cw + row * DX_COLS + (((2 * ch) ^ (row & 7)) << 2));
        const f32x4 hi = *(const f32x4*)(cw + row * DX_COLS + (((2 * ch + 1) ^ (row & 7)) << 2));
        if (m < a.M) {
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (a.R) {
                const bf16x8 r = *(const bf16x8*)(a.R + (long)m * a.ldr + n0 + ch * 8);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += (float)r[q];
            }
            store_f<8>(a.out + (long)m * a.ldo + n0 + ch * 8, v);
        }
    }
}

}  // namespace

bool av_lora_dx_masked_supported(int dtype, int N, int r, const long* ldt, const long* ldat, int nj, long ldr, long ldo) {
    if (dtype != AV_BF16 || N % DX_COLS != 0 || r > 32 || nj < 1 || nj > 3 || ldr % 8 != 0 || ldo % 8 != 0) return false;
    for (int j = 0; j < nj; ++j)
        if (ldt[j] % 8 != 0 || ldat[j] % 8 != 0 || ldt[j] < 32 || ldat[j] < 32) return false;
    return true;
}

int av_lora_dx_masked(const void* const* T, const long* ldt, const void* const* AT, const long* ldat, const uint32_t* seeds, int nj, int r,
                      const void* R, long ldr, void* out, long ldo, int M, int N, float p, const uint32_t* seed_dev, int dtype, hipStream_t st) {
    AV_CHECK_ARG(T && AT && ldt && ldat && seeds && out && M > 0 && N > 0, "lora_dx_masked: null/empty");
    AV_CHECK_ARG(av_lora_dx_masked_supported(dtype, N, r, ldt, ldat, nj, R ? ldr : 8, ldo),
                 "lora_dx_masked: bf16 only, N %% 128 == 0 (N=%d), rank <= 32 (r=%d), 1..3 adapters (nj=%d), 16-byte rows", N, r, nj);
    AV_CHECK_ARG(p >= 0.f && p < 1.f, "lora_dx_masked: p=%f", p);
    LoraDxArgs a = {};
    for (int j = 0; j < nj; ++j) {
        AV_CHECK_ARG(T[j] && AT[j], "lora_dx_masked: null adapter %d", j);
        a.T[j] = (const bf16*)T[j]; a.AT[j] = (const bf16*)AT[j]; a.ldt[j] = ldt[j]; a.ldat[j] = ldat[j]; a.seed[j] = seeds[j];
    }
    a.R = (const bf16*)R; a.out = (bf16*)out; a.ldr = ldr; a.ldo = ldo; a.M = M; a.N = N; a.nj = nj; a.p = p; a.seed_dev = seed_dev;
    const dim3 grid(N / DX_COLS, av_cdiv(M, 4 * DX_ROWS));
    if (nj == 1) hipLaunchKernelGGL(lora_dx_masked_kernel<1>, grid, dim3(256), 0, st, a);
    else if (nj == 2) hipLaunchKernelGGL(lora_dx_masked_kernel<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(lora_dx_masked_kernel<3>, grid, dim3(256), 0, st, a);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int avllm_lora_dx_masked(const void* const* T, const int64_t* ldt, const void* const* AT, const int64_t* ldat, const uint32_t* seeds,
                                    int32_t nj, int32_t r, const void* R, int64_t ldr, void* out, int64_t ldo, int32_t M, int32_t N, float p,
                                    const uint32_t* seed_dev, int32_t dtype, void* stream) {
    AV_CHECK_ARG(nj >= 1 && nj <= 3 && ldt && ldat, "lora_dx_masked: nj=%d", nj);
    long lt[3], la[3];
    for (int j = 0; j < nj; ++j) { lt[j] = (long)ldt[j]; la[j] = (long)ldat[j]; }
    return av_lora_dx_masked(T, lt, AT, la, seeds, nj, r, R, ldr, out, ldo, M, N, p, seed_dev, dtype, (hipStream_t)stream);
}
