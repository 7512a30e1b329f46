// NT GEMM for gfx950:  C[M,N] = act(alpha * (A[M,K] . B[N,K]^T + A2[M,K2] . B2[N,K2]^T) + bias[N]) + R
//
// Every dense contraction of the hot path is this one shape (activations x frozen weights stored
// [out,in], exactly as nn.Linear holds them -- reference call sites: whisper/clip/llama projections,
// SURVEY.md §2 "Stock ops").  The optional second K segment carries the LoRA up-projection
// (A2 = x.A^T padded to 64 columns, B2 = lora_B padded to 64 columns) so the adapter costs one extra
// K-tile inside the same accumulator instead of a second pass over C (peft lora.Linear; reference wrap
// src/clip_whisper/models/clip_whisper_model.py:961-1005).
//
// bf16 kernel: 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 accumulators.
//   * both operands staged global->LDS by `global_load_lds_dwordx4` (no VGPR round trip), two LDS
//     buffers, one barrier per K-tile; LDS image is lane-linear, bank conflicts removed by XOR-ing the
//     16-B chunk index with (row & 7) on the SOURCE address and again on the ds_read_b128 address.
//   * MFMA is issued with the weight fragment as the A operand (acc = W_frag x X_frag), so each lane ends
//     with 4 consecutive output columns of one row -> one 8-byte bf16 store per accumulator.
//   * XCD-aware block->tile map: the 8 XCDs get contiguous tile ranges so that neighbours share panels in L2.
// f32 kernel (strict-parity mode): 64x64x16 tile on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
#include "common.h"
#include "avllm_internal.h"
#include "gemm_shared.h"
#include <stdlib.h>
#include <utility>
#include <type_traits>

namespace {

struct EpiParams {
    void* C; long ldc; int out_f32;
    const void* bias; const void* R; long ldr; int r_mod;
    int g_in, g_out, g_off;
    float alpha; int act;
    int M, N;
    uint32_t drop_seed; float drop_p; const uint32_t* seed_dev;
};

// v[4] = 4 consecutive output columns n0..n0+3 of logical row m
template <typename T>
__device__ __forceinline__ void epilogue_store4(const EpiParams& e, int m, int n0, float (&v)[4]) {
    if (m >= e.M || n0 >= e.N) return;
    const bool full = (n0 + 3 < e.N);
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (e.bias) {
        const T* bp = (const T*)e.bias + n0;
        if (full) { load_f<4>(bp, b); } else { for (int i = 0; i < 4 && n0 + i < e.N; ++i) b[i] = to_f(bp[i]); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = act_apply(v[i] * e.alpha + b[i], e.act);
    if (e.drop_p > 0.f) {
        const float sc = av_drop_scale(e.drop_p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = av_keep(av_seed(e.seed_dev, e.drop_seed), (unsigned long long)m * e.N + n0 + i, e.drop_p) ? v[i] * sc : 0.f;
    }
    if (e.R) {
        const long rr = e.r_mod > 0 ? (m % e.r_mod) : m;
        const T* rp = (const T*)e.R + rr * e.ldr + n0;
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (full) { load_f<4>(rp, r); } else { for (int i = 0; i < 4 && n0 + i < e.N; ++i) r[i] = to_f(rp[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += r[i];
    }
    long orow = m;
    if (e.g_in > 0) orow = (long)(m / e.g_in) * e.g_out + e.g_off + (m % e.g_in);
    if (e.out_f32) {
        float* cp = (float*)e.C + orow * e.ldc + n0;
        if (full) store_f<4>(cp, v); else for (int i = 0; i < 4 && n0 + i < e.N; ++i) cp[i] = v[i];
    } else {
        T* cp = (T*)e.C + orow * e.ldc + n0;
        if (full) store_f<4>(cp, v); else for (int i = 0; i < 4 && n0 + i < e.N; ++i) cp[i] = from_f<T>(v[i]);
    }
}

// v[8] = 8 consecutive output columns of logical row m; every pointer is 16-byte aligned on this path (checked at launch).
// Single rounding: bias/activation/dropout/residual are applied to the fp32 accumulator, then one convert + one 16-byte store.
__device__ __forceinline__ void epilogue_store8(const EpiParams& e, int m, int n0, float (&v)[8]) {
    if (e.bias) {
        float b[8];
        load_f<8>((const bf16*)e.bias + n0, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * e.alpha + b[i];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
    }
    if (e.act != AV_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = act_apply_fast(v[i], e.act);
    }
    if (e.drop_p > 0.f) {
        av_mask8(v, av_seed(e.seed_dev, e.drop_seed), (unsigned long long)m * e.N + n0, av_drop_thr(e.drop_p), av_drop_scale(e.drop_p));      // N % 8 == 0: even start
    }
    if (e.R) {
        const long rr = e.r_mod > 0 ? (m % e.r_mod) : m;
        float r[8];
        load_f<8>((const bf16*)e.R + rr * e.ldr + n0, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += r[i];
    }
    long orow = m;
    if (e.g_in > 0) orow = (long)(m / e.g_in) * e.g_out + e.g_off + (m % e.g_in);
    if (e.out_f32) store_f<8>((float*)e.C + orow * e.ldc + n0, v);
    else store_f<8>((bf16*)e.C + orow * e.ldc + n0, v);
}

using avg::epilogue_fast8; using avg::xcd_remap; using avg::tile_coords;

// cache-policy bits of the operand DMA of the 4-wave kernels (experiment: tools/gemm_dma_policy.sh builds one library per policy); default none
#ifndef AVLLM_DMA_POLICY
#define AVLLM_DMA_POLICY ""
#endif

// ------------------------------------------------------------------------------------------ bf16
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand per stage

struct GemmArgs {
    const bf16* A; const bf16* B; const bf16* A2; const bf16* B2;
    long lda, ldb, lda2, ldb2;
    int K, K2;
    int wide_epi;        // outputs/bias/residual 16-byte aligned, N % 8 == 0: LDS-staged epilogue with 16-byte row-coalesced stores
    int gw;              // persistent kernel, tall shapes: tile-column group width of the walk (gemm_shared.h tile_coords); 0 = plain row-major
    int dbg;             // experiment knobs of the persistent kernel (env AVLLM_GEMM_DBG): bit 0 = no epilogue stores, bit 1 = strict first-K-step wait, bit 2 = row-major tile order instead of 8 x 4 blocks, bit 3 = per-XCD contiguous id ranges instead of one block per XCD and round, bits 4.. = start-stagger phases (1 = off)
    EpiParams e;
};

__device__ __forceinline__ void stage_tile(const bf16* __restrict__ base, long ld, int row0, int rows_max,
                                           int k0, char* lds, int wave, int lane) {
    // one wave-instruction = 8 rows x 128 B.  4 waves x 4 passes = 128 rows.
    const int rsub = lane >> 3;                       // row inside the 8-row group == (row & 7)
    const int chunk_src = (lane & 7) ^ rsub;          // inverse swizzle on the source
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = p * 32 + wave * 8 + rsub;
        int gr = row0 + r;
        gr = gr < rows_max ? gr : rows_max - 1;       // clamp: rows past the edge are never stored
        const bf16* src = base + (long)gr * ld + k0 + chunk_src * 8;
        char* dst = lds + (p * 32 + wave * 8) * 128;  // wave-uniform base; hardware adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (A tile + B tile) = 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (g.e.M + BM - 1) / BM, tiles_n = (g.e.N + BN - 1) / BN;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt1 = g.K / BK, nt = nt1 + g.K2 / BK;
    auto stage = [&](int t, int buf) {
        char* a_lds = smem + buf * (2 * TILE_BYTES);
        char* b_lds = a_lds + TILE_BYTES;
        if (t < nt1) {
            stage_tile(g.A, g.lda, m0, g.e.M, t * BK, a_lds, wave, lane);
            stage_tile(g.B, g.ldb, n0, g.e.N, t * BK, b_lds, wave, lane);
        } else {
            stage_tile(g.A2, g.lda2, m0, g.e.M, (t - nt1) * BK, a_lds, wave, lane);
            stage_tile(g.B2, g.ldb2, n0, g.e.N, (t - nt1) * BK, b_lds, wave, lane);
        }
    };

    stage(0, 0);
    __syncthreads();                                   // hipcc drains vmcnt(0) before the barrier
    const int fr = lane & 15, fq = lane >> 4;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) stage(t + 1, cur ^ 1);
        const char* a_lds = smem + cur * (2 * TILE_BYTES);
        const char* b_lds = a_lds + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xa[4], wb[4];
            const int chunk = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wm * 64 + i * 16 + fr;
                xa[i] = *(const bf16x8*)(a_lds + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wn * 64 + j * 16 + fr;
                wb[j] = *(const bf16x8*)(b_lds + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // acc[i][j][reg]: output row m = m0 + wm*64 + i*16 + (lane&15); column n = n0 + wn*64 + j*16 + (lane>>4)*4 + reg
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            epilogue_store4<bf16>(g.e, m, n0 + wn * 64 + j * 16 + fq * 4, v);
        }
    }
}

// ------------------------------------------------------------------------------------------ bf16, 256x128 tile, 3-stage ring
// Large-M variant: 8 waves (4x2, 64x64 each) share a 256x128 tile; a 3-deep LDS ring (3 x 48 KiB) keeps TWO K-tiles of
// global_load_lds in flight across the single raw s_barrier per K-tile (counted s_waitcnt vmcnt(6), never 0 in the loop),
// which is what the 2-buffer kernel above cannot do: its __syncthreads drains vmcnt(0) every K-tile.
constexpr int LBM = 256, LBN = 128;
constexpr int LSTAGE = (LBM + LBN) * BK * 2;      // 48 KiB
constexpr int LNSTAGE = 3;

__device__ __forceinline__ void stage_rows8(const bf16* __restrict__ base, long ld, int row0, int rows_max, int k0, char* lds,
                                            int group, int lane) {
    // one wave-instruction: 8 rows x 128 B at LDS rows [group*8, group*8+8)
    const int rsub = lane >> 3;
    int gr = row0 + group * 8 + rsub;
    gr = gr < rows_max ? gr : rows_max - 1;
    const bf16* src = base + (long)gr * ld + k0 + (((lane & 7) ^ rsub) << 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + group * 1024), 16, 0, 0);
}

__global__ __launch_bounds__(512, 2) void gemm_bf16_l_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (g.e.M + LBM - 1) / LBM, tiles_n = (g.e.N + LBN - 1) / LBN;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * LBM, n0 = tn * LBN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt1 = g.K / BK, nt = nt1 + g.K2 / BK;
    auto stage = [&](int t, int buf) {
        char* a_lds = smem + buf * LSTAGE;
        char* b_lds = a_lds + LBM * BK * 2;
        const bf16* Ap = t < nt1 ? g.A : g.A2;
        const bf16* Bp = t < nt1 ? g.B : g.B2;
        const long la = t < nt1 ? g.lda : g.lda2, lb = t < nt1 ? g.ldb : g.ldb2;
        const int k0 = (t < nt1 ? t : t - nt1) * BK;
#pragma unroll
        for (int p = 0; p < 4; ++p) stage_rows8(Ap, la, m0, g.e.M, k0, a_lds, p * 8 + wave, lane);
#pragma unroll
        for (int p = 0; p < 2; ++p) stage_rows8(Bp, lb, n0, g.e.N, k0, b_lds, p * 8 + wave, lane);
    };

    stage(0, 0);
    if (nt > 1) stage(1, 1);
    const int fr = lane & 15, fq = lane >> 4;
    int buf = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 2 < nt) stage(t + 2, buf >= 1 ? buf - 1 : LNSTAGE - 1);      // (t+2) % 3 == (buf+2) % 3
        const char* a_lds = smem + buf * LSTAGE;
        const char* b_lds = a_lds + LBM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xa[4], wb[4];
            const int chunk = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wm * 64 + i * 16 + fr;
                xa[i] = *(const bf16x8*)(a_lds + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wn * 64 + j * 16 + fr;
                wb[j] = *(const bf16x8*)(b_lds + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
        buf = buf + 1 == LNSTAGE ? 0 : buf + 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            epilogue_store4<bf16>(g.e, m, n0 + wn * 64 + j * 16 + fq * 4, v);
        }
    }
}

// ------------------------------------------------------------------------------------------ bf16, 256x256 tile, 16 waves x (64x64)
// tools/ubench/gemm_ablate: at 128x128 the LDS-DMA stream alone (18.8 TB/s chip-wide, ~35 B/clk/CU) takes as long as the
// ds_read+MFMA phase alone, so the tile is load-bound at ~56 % of MFMA peak.  This variant keeps the per-wave code of the
// 128x128 kernel (64x64 per wave, its LDS-fed phase sustains ~1.5 PF/s) but lets 16 waves (4x4, 1024 threads, 4 per SIMD)
// share one 256x256 tile: half the global->LDS bytes per FLOP.  Two 64 KiB stages = 128 KiB LDS, one workgroup per CU.
constexpr int HBM_ = 256, HBN_ = 256;
constexpr int HSTAGE = (HBM_ + HBN_) * BK * 2;     // 64 KiB

__global__ __launch_bounds__(1024, 4) void gemm_bf16_h_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_m = (g.e.M + HBM_ - 1) / HBM_, tiles_n = (g.e.N + HBN_ - 1) / HBN_;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * HBM_, n0 = tn * HBN_;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt1 = g.K / BK, nt = nt1 + g.K2 / BK;
    // Staging addresses: this thread's four loads (row groups wave and 16+wave of A and of B) keep their element offsets
    // row*ld + swizzled chunk in registers for both K segments, so a K-tile costs one 64-bit add per load instead of a
    // 64-bit multiply chain (32-bit offsets: the dispatcher only sends operands below 2^32 elements here).
    const int rsub = lane >> 3, swz = ((lane & 7) ^ rsub) << 3;
    unsigned offA0[2], offA1[2], offB0[2], offB1[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        int ra = m0 + (p * 16 + wave) * 8 + rsub, rb = n0 + (p * 16 + wave) * 8 + rsub;
        ra = ra < g.e.M ? ra : g.e.M - 1;                     // rows past the edge are never stored
        rb = rb < g.e.N ? rb : g.e.N - 1;
        offA0[p] = (unsigned)ra * (unsigned)g.lda + swz; offA1[p] = (unsigned)ra * (unsigned)g.lda2 + swz;
        offB0[p] = (unsigned)rb * (unsigned)g.ldb + swz; offB1[p] = (unsigned)rb * (unsigned)g.ldb2 + swz;
    }
    auto stage = [&](int t, int buf) {
        char* a_lds = smem + buf * HSTAGE;
        char* b_lds = a_lds + HBM_ * BK * 2;
        const bool seg2 = t >= nt1;
        const bf16* Ap = (seg2 ? g.A2 : g.A) + (seg2 ? t - nt1 : t) * BK;
        const bf16* Bp = (seg2 ? g.B2 : g.B) + (seg2 ? t - nt1 : t) * BK;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ap + (seg2 ? offA1[p] : offA0[p])),
                                             (__attribute__((address_space(3))) void*)(a_lds + (p * 16 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bp + (seg2 ? offB1[p] : offB0[p])),
                                             (__attribute__((address_space(3))) void*)(b_lds + (p * 16 + wave) * 1024), 16, 0, 0);
    };

    stage(0, 0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) stage(t + 1, cur ^ 1);
        const char* a_lds = smem + cur * HSTAGE;
        const char* b_lds = a_lds + HBM_ * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xa[4], wb[4];
            const int chunk = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wm * 64 + i * 16 + fr;
                xa[i] = *(const bf16x8*)(a_lds + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wn * 64 + j * 16 + fr;
                wb[j] = *(const bf16x8*)(b_lds + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if (g.wide_epi) {
        // Epilogue through LDS (both stages are free after the last barrier).  Per-lane 8-byte stores of the MFMA layout are
        // store-ISSUE-bound (~7 B/clk/CU: 128 KiB of output cost ~9 us per tile, 30 % of a K=768 tile); here each half tile goes
        // to LDS as fp32 and comes back as whole 16-byte row chunks: bias/residual loads and the stores are row-coalesced.
        constexpr int CT_LD = 256;                           // floats per LDS row; 16-byte chunk index XOR (row & 7) instead of padding
        float* ct = (float*)smem;                            // [128][256] fp32 = 131,072 B = 2 * HSTAGE exactly
        if (g.e.alpha == 1.f && g.e.drop_p <= 0.f && g.e.g_in <= 0 && !g.e.out_f32 && g.e.r_mod <= 0 && m0 + HBM_ <= g.e.M && n0 + HBN_ <= g.e.N) {
            // thread = one 8-column chunk (tid & 31) of rows (tid >> 5) + 32 p: its bias, its LDS chunk pair and its row strides are fixed
            const int ch = tid & 31, prow = tid >> 5;
            float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (g.e.bias) load_f<8>((const bf16*)g.e.bias + n0 + ch * 8, b);
            const float* clo = ct + prow * CT_LD + (((2 * ch) ^ (prow & 7)) << 2);
            const float* chi = ct + prow * CT_LD + (((2 * ch + 1) ^ (prow & 7)) << 2);
            bf16* cp = (bf16*)g.e.C + (long)(m0 + prow) * g.e.ldc + n0 + ch * 8;
            const bf16* rp = g.e.R ? (const bf16*)g.e.R + (long)(m0 + prow) * g.e.ldr + n0 + ch * 8 : nullptr;
            const long cstep = 32 * g.e.ldc, rstep = g.e.R ? 32 * g.e.ldr : 0;
            auto run = [&](auto actc) __attribute__((always_inline)) {
                constexpr int ACT = decltype(actc)::value;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if ((wm >> 1) == half) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int row = (wm & 1) * 64 + i * 16 + fr;
                                *(f32x4*)(ct + row * CT_LD + (((wn * 16 + j * 4 + fq) ^ (row & 7)) << 2)) = acc[i][j];
                            }
                    }
                    __syncthreads();
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        epilogue_fast8<ACT>(*(const f32x4*)(clo + p * 32 * CT_LD), *(const f32x4*)(chi + p * 32 * CT_LD), b, g.e.bias != nullptr, rp, cp);
                        cp += cstep; rp += rstep;
                    }
                    __syncthreads();
                }
            };
            if (g.e.act == AV_ACT_NONE) run(std::integral_constant<int, AV_ACT_NONE>{});
            else if (g.e.act == AV_ACT_GELU) run(std::integral_constant<int, AV_ACT_GELU>{});
            else if (g.e.act == AV_ACT_QUICK_GELU) run(std::integral_constant<int, AV_ACT_QUICK_GELU>{});
            else run(std::integral_constant<int, AV_ACT_SILU>{});
            return;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((wm >> 1) == half) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                    {
                        const int row = (wm & 1) * 64 + i * 16 + fr;
                        *(f32x4*)(ct + row * CT_LD + (((wn * 16 + j * 4 + fq) ^ (row & 7)) << 2)) = acc[i][j];
                    }
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int idx = p * 1024 + tid, row = idx >> 5, ch = idx & 31;
                const int m = m0 + half * 128 + row, n = n0 + ch * 8;
                if (m < g.e.M && n < g.e.N) {
                    const f32x4 lo = *(const f32x4*)(ct + row * CT_LD + (((2 * ch) ^ (row & 7)) << 2));
                    const f32x4 hi = *(const f32x4*)(ct + row * CT_LD + (((2 * ch + 1) ^ (row & 7)) << 2));
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    epilogue_store8(g.e, m, n, v);
                }
            }
            __syncthreads();
        }
        return;
    }
#pragma clang loop unroll(full)
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
#pragma clang loop unroll(full)
        for (int j = 0; j < 4; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            epilogue_store4<bf16>(g.e, m, n0 + wn * 64 + j * 16 + fq * 4, v);
        }
    }
}

// ------------------------------------------------------------------------------------------ bf16, 256x256 tile, 4 waves x (128x128)
// One wave per SIMD, each owning a 128x128 quadrant: 0.25 ds_read_b128 per MFMA instead of the 0.5 of a 64x64 wave tile
// (at 16 waves the LDS port, reads + DMA writes, is busier than the MFMA pipe).  A K-step is 64 wide = 128 MFMAs per wave; the
// fragments of BOTH k-halves of both operands sit in registers (128 VGPRs, accumulators in the 256 AGPRs), so an operand's half
// of the current LDS buffer is dead after 8 ds_reads per wave and is refilled IN PLACE by the global->LDS DMA with the data of
// K-step t+2: two 64-KiB buffers give two K-steps of load latency cover.  The loop is written instruction by instruction (the
// compiler's own schedule for a single-wave-per-SIMD loop moved accumulators between AGPRs and VGPRs every iteration): three
// barriers and three full lgkmcnt waits per 128 MFMAs, every ds_read issued >= 10 MFMAs before its wait, loads one per 2-3
// MFMAs with SGPR base + 32-bit tile-relative VGPR offsets (no 64-bit address arithmetic, no operand-size limit).
// tools/ubench/gemm_pingpong.hip g4h: 1.25-1.40 PF/s at 4096x4096x11008 against 1.15-1.17 for the 16-wave kernel in the same run.
template <int N, int BUF>
__device__ __forceinline__ void wgemm_step(f32x4 (&acc)[8][8], bf16x8 (&FA)[2][8], bf16x8 (&FB)[2][8], const int (&la)[2][2], const int (&lb)[2][2],
                                           const unsigned (&vA)[8], const unsigned (&vB)[8], const bf16* pA, const bf16* pB, int m0A, int m0B) {
    constexpr int h = N >> 6, n = N & 63, I = n >> 3, J = n & 7;
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(FB[h][J]), "v"(FA[h][I]));
#define AV_W_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define AV_W_LD(voff, base, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" AVLLM_DMA_POLICY :: "s"(m0v), "v"(voff), "s"(base) : "memory")
    if constexpr (h == 0) {
        if constexpr (n < 16 && (n & 1)) AV_W_RD(FB[1][n >> 1], lb[BUF][1], (n >> 1) * 2048);
        if constexpr (n == 20) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (n == 21) __builtin_amdgcn_s_barrier();                          // every wave has read all of B(cur)
        if constexpr (n >= 22 && n < 38 && !(n & 1)) AV_W_LD(vB[(n - 22) >> 1], pB, m0B + ((n - 22) >> 1) * 1024);
        if constexpr (n >= 23 && n < 39 && (n & 1)) AV_W_RD(FA[1][(n - 23) >> 1], la[BUF][1], ((n - 23) >> 1) * 2048);
        if constexpr (n == 50) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (n == 51) __builtin_amdgcn_s_barrier();                          // ... and all of A(cur)
        if constexpr (n == 52 || n == 55 || n == 58 || n == 61) AV_W_LD(vA[(n - 52) / 3], pA, m0A + ((n - 52) / 3) * 1024);
    } else {
        if constexpr (n == 0) AV_W_LD(vA[4], pA, m0A + 4 * 1024);
        if constexpr (n == 26) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");      // 13 loads of this step issued so far: all older ones have landed
        if constexpr (n == 27) __builtin_amdgcn_s_barrier();                          // the other buffer is complete for every wave
        if constexpr (n >= 28 && n < 36) AV_W_RD(FB[0][n - 28], lb[BUF ^ 1][0], (n - 28) * 2048);
        if constexpr (n >= 37 && n < 53 && (n & 1)) AV_W_RD(FA[0][(n - 37) >> 1], la[BUF ^ 1][0], ((n - 37) >> 1) * 2048);
        if constexpr (n == 32 || n == 40 || n == 48) AV_W_LD(vA[5 + (n - 32) / 8], pA, m0A + (5 + (n - 32) / 8) * 1024);
        if constexpr (n == 62) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int BUF, int... Ns>
__device__ __forceinline__ void wgemm_kstep(std::integer_sequence<int, Ns...>, f32x4 (&acc)[8][8], bf16x8 (&FA)[2][8], bf16x8 (&FB)[2][8], const int (&la)[2][2],
                                            const int (&lb)[2][2], const unsigned (&vA)[8], const unsigned (&vB)[8], const bf16* pA, const bf16* pB, int m0A, int m0B) {
    (wgemm_step<Ns, BUF>(acc, FA, FB, la, lb, vA, vB, pA, pB, m0A, m0B), ...);
}

__global__ __launch_bounds__(256, 1) void gemm_bf16_w_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = (g.e.M + HBM_ - 1) / HBM_, tiles_n = (g.e.N + HBN_ - 1) / HBN_;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * HBM_, n0 = tn * HBN_;
    const int nt1 = g.K / BK, nt = nt1 + g.K2 / BK;                  // >= 2 (dispatcher)

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LDS: buffer b at b*64 KiB = A rows 0..255 then B rows 0..255, 128 B per row, 16-byte chunk index ^ (row & 7)
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    int la[2][2], lb[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pc = ((4 * h + fq) ^ (fr & 7)) << 4;
            la[b][h] = lds0 + b * HSTAGE + (wr * 128 + fr) * 128 + pc;
            lb[b][h] = lds0 + b * HSTAGE + HBM_ * 128 + (wc * 128 + fr) * 128 + pc;
        }
    // Staging: one load = 8 rows x 128 B; this wave owns rows wave*64 + q*8 + (lane >> 3) of each operand.  Byte offsets relative to the
    // tile's first row, per K segment (rows past the edge are clamped: they are never stored)
    unsigned vA1[8], vB1[8], vA2[8], vB2[8];
    const int ch = ((lane & 7) ^ (lane >> 3)) << 4;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = wave * 64 + q * 8 + (lane >> 3);
        const unsigned ra = m0 + r < g.e.M ? r : g.e.M - 1 - m0, rb = n0 + r < g.e.N ? r : g.e.N - 1 - n0;
        vA1[q] = ra * (unsigned)g.lda * 2 + ch; vA2[q] = ra * (unsigned)g.lda2 * 2 + ch;
        vB1[q] = rb * (unsigned)g.ldb * 2 + ch; vB2[q] = rb * (unsigned)g.ldb2 * 2 + ch;
    }
    const bf16* tA1 = g.A + (long)m0 * g.lda; const bf16* tB1 = g.B + (long)n0 * g.ldb;
    const bf16* tA2 = g.K2 ? g.A2 + (long)m0 * g.lda2 : g.A; const bf16* tB2 = g.K2 ? g.B2 + (long)n0 * g.ldb2 : g.B;
    const int mw = lds0 + wave * 8192;                              // this wave's 64 rows inside an operand region
    auto stage = [&](int kt, int buf) {
        const bool s2 = kt >= nt1;
        const bf16* pa = s2 ? tA2 + (long)(kt - nt1) * BK : tA1 + (long)kt * BK;
        const bf16* pb = s2 ? tB2 + (long)(kt - nt1) * BK : tB1 + (long)kt * BK;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            AV_W_LD(s2 ? vA2[q] : vA1[q], pa, mw + buf * HSTAGE + q * 1024);
            AV_W_LD(s2 ? vB2[q] : vB1[q], pb, mw + buf * HSTAGE + HBM_ * 128 + q * 1024);
        }
    };
    bf16x8 FA[2][8], FB[2][8];
    stage(0, 0); stage(1, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) { AV_W_RD(FB[0][j], lb[0][0], 0); lb[0][0] += 2048; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { AV_W_RD(FA[0][i], la[0][0], 0); la[0][0] += 2048; }
    lb[0][0] -= 8 * 2048; la[0][0] -= 8 * 2048;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    using Seq = std::make_integer_sequence<int, 128>;
    // K-step t (buffer t & 1) issues the loads of K-step t+2 into its own buffer; past the end the last K-tile is loaded again into a
    // buffer nobody reads any more, which keeps the vmcnt arithmetic of the loop uniform
    auto kstep = [&](int t, auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        const int kl = t + 2 < nt ? t + 2 : nt - 1;
        const bool s2 = kl >= nt1;
        const bf16* pa = s2 ? tA2 + (long)(kl - nt1) * BK : tA1 + (long)kl * BK;
        const bf16* pb = s2 ? tB2 + (long)(kl - nt1) * BK : tB1 + (long)kl * BK;
        unsigned va[8], vb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { va[q] = s2 ? vA2[q] : vA1[q]; vb[q] = s2 ? vB2[q] : vB1[q]; }
        wgemm_kstep<BUF>(Seq{}, acc, FA, FB, la, lb, va, vb, pa, pb, mw + BUF * HSTAGE, mw + BUF * HSTAGE + HBM_ * 128);
    };
    int t = 0;
    for (; t + 1 < nt; t += 2) { kstep(t, std::integral_constant<int, 0>{}); kstep(t + 1, std::integral_constant<int, 1>{}); }
    if (t < nt) kstep(t, std::integral_constant<int, 0>{});
    // The DMA writes of the dummy loads must have landed before LDS is reused.  The nops cover the result latency of the last MFMAs, which
    // the compiler's hazard recogniser cannot see through inline asm; naming every accumulator as an in/out operand of these statements
    // keeps compiler-generated readers (register copies, spill stores) behind them -- without that hipcc placed a scratch store of the
    // last accumulator directly after its MFMA and spilled a stale value.
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#define AV_W_PIN8(I) asm volatile("" : "+a"(acc[I][0]), "+a"(acc[I][1]), "+a"(acc[I][2]), "+a"(acc[I][3]), "+a"(acc[I][4]), "+a"(acc[I][5]), "+a"(acc[I][6]), "+a"(acc[I][7]))
    AV_W_PIN8(0); AV_W_PIN8(1); AV_W_PIN8(2); AV_W_PIN8(3); AV_W_PIN8(4); AV_W_PIN8(5); AV_W_PIN8(6); AV_W_PIN8(7);
#undef AV_W_PIN8
    __syncthreads();
#undef AV_W_RD
#undef AV_W_LD
    if (g.wide_epi) {
        // Epilogue through LDS, as in the 16-wave kernel: each 128-row half goes to LDS as fp32 and comes back as whole 16-byte row chunks
        constexpr int CT_LD = 256;
        float* ct = (float*)smem;                                    // [128][256] fp32 = both stage buffers
        if (g.e.alpha == 1.f && g.e.drop_p <= 0.f && g.e.g_in <= 0 && !g.e.out_f32 && g.e.r_mod <= 0 && m0 + HBM_ <= g.e.M && n0 + HBN_ <= g.e.N) {
            // lean items (see the 16-wave kernel): thread = 8-column chunk (tid & 31) of rows (tid >> 5) + 8 p
            const int ch = tid & 31, prow = tid >> 5;
            float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (g.e.bias) load_f<8>((const bf16*)g.e.bias + n0 + ch * 8, b);
            const float* clo = ct + prow * CT_LD + (((2 * ch) ^ prow) << 2);
            const float* chi = ct + prow * CT_LD + (((2 * ch + 1) ^ prow) << 2);
            bf16* cp = (bf16*)g.e.C + (long)(m0 + prow) * g.e.ldc + n0 + ch * 8;
            const bf16* rp = g.e.R ? (const bf16*)g.e.R + (long)(m0 + prow) * g.e.ldr + n0 + ch * 8 : nullptr;
            const long cstep = 8 * g.e.ldc, rstep = g.e.R ? 8 * g.e.ldr : 0;
            auto run = [&](auto actc) __attribute__((always_inline)) {
                constexpr int ACT = decltype(actc)::value;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (wr == half) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int row = i * 16 + fr;
                                *(f32x4*)(ct + row * CT_LD + (((wc * 32 + j * 4 + fq) ^ (row & 7)) << 2)) = acc[i][j];
                            }
                    }
                    __syncthreads();
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        epilogue_fast8<ACT>(*(const f32x4*)(clo + p * 8 * CT_LD), *(const f32x4*)(chi + p * 8 * CT_LD), b, g.e.bias != nullptr, rp, cp);
                        cp += cstep; rp += rstep;
                    }
                    __syncthreads();
                }
            };
            if (g.e.act == AV_ACT_NONE) run(std::integral_constant<int, AV_ACT_NONE>{});
            else if (g.e.act == AV_ACT_GELU) run(std::integral_constant<int, AV_ACT_GELU>{});
            else if (g.e.act == AV_ACT_QUICK_GELU) run(std::integral_constant<int, AV_ACT_QUICK_GELU>{});
            else run(std::integral_constant<int, AV_ACT_SILU>{});
            return;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (wr == half) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int row = i * 16 + fr;
                        *(f32x4*)(ct + row * CT_LD + (((wc * 32 + j * 4 + fq) ^ (row & 7)) << 2)) = acc[i][j];
                    }
            }
            __syncthreads();
#pragma unroll 4
            for (int p = 0; p < 16; ++p) {
                const int idx = p * 256 + tid, row = idx >> 5, c8 = idx & 31;
                const int m = m0 + half * 128 + row, n = n0 + c8 * 8;
                if (m < g.e.M && n < g.e.N) {
                    const f32x4 lo = *(const f32x4*)(ct + row * CT_LD + (((2 * c8) ^ (row & 7)) << 2));
                    const f32x4 hi = *(const f32x4*)(ct + row * CT_LD + (((2 * c8 + 1) ^ (row & 7)) << 2));
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    epilogue_store8(g.e, m, n, v);
                }
            }
            __syncthreads();
        }
        return;
    }
#pragma clang loop unroll(full)
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + fr;
#pragma clang loop unroll(full)
        for (int j = 0; j < 8; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            epilogue_store4<bf16>(g.e, m, n0 + wc * 128 + j * 16 + fq * 4, v);
        }
    }
}

// Persistent form of the 4-wave kernel for calls the lean epilogue covers (bf16 output, alpha 1, no dropout / row remap / broadcast
// residual).  One workgroup per CU walks tiles vid, vid+G, ...; the K-step pipeline never drains between them: the last two K-steps of a
// tile issue the loads of the NEXT tile's first two K-steps (where the one-shot kernel issues dummy loads), so a tile's prologue latency
// and the workgroup relaunch disappear behind the previous tile.  The epilogue therefore stays out of the two stage buffers: each wave
// transposes its own 128x128 quadrant 16 rows at a time through a private 8-KiB slice of the remaining 32 KiB of LDS (same-wave LDS
// traffic is in order: no workgroup barrier), and the accumulators are re-initialised for free by the first K-step's MFMAs taking 0 as
// their C operand.  Buffers alternate by address arithmetic, so any K-step count and tile parity runs the same loop body.
// B-operand (weight) fragment j of a wave reads LDS rows 32 (j >> 1) + 4 (j & 1) + 8 (fr >> 2) + (fr & 3) of its 128-row half: MFMA tiles 2p and
// 2p+1 then hold, in lane (fr, fq), output columns 32p + 8fq + {0..3} and + {4..7} of row fr -- one 16-byte bf16 chunk per lane and tile
// pair, stored straight from the accumulators (no LDS transpose in the epilogue).  Byte offset of fragment j from the lane's base row:
#define WP_BOFF(j) ((((j) >> 1) * 32 + ((j) & 1) * 4) * 128)
// MODE 0: any K-step but a tile's first; 1: a tile's first K-step (accumulators start from 0); 2: the same right after the epilogue of a
// tile that lay fully inside the matrix.  vmcnt counts loads and stores together in issue order, so the counted wait of the K-step would
// also wait for the 32 epilogue stores issued just before it (their write acknowledgements take several microseconds when the whole
// grid stores at once); mode 2 allows exactly those 32 to stay in flight: everything OLDER (the K-step's operands) has still landed.
template <int N, int MODE>
__device__ __forceinline__ void wpgemm_step(f32x4 (&acc)[8][8], bf16x8 (&FA)[2][8], bf16x8 (&FB)[2][8], const int (&la)[2], const int (&lb)[2],
                                            const unsigned (&vA)[8], const unsigned (&vB)[8], const bf16* pA, const bf16* pB, int m0A, int m0B) {
    constexpr int h = N >> 6, n = N & 63, I = n >> 3, J = n & 7;
    constexpr bool FIRST = MODE != 0;
    if constexpr (FIRST && h == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[I][J]) : "v"(FB[h][J]), "v"(FA[h][I]));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(FB[h][J]), "v"(FA[h][I]));
#define AV_W_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define AV_W_LD(voff, base, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" AVLLM_DMA_POLICY :: "s"(m0v), "v"(voff), "s"(base) : "memory")
    // la/lb[1]: this lane's k-half-1 fragment address in the CURRENT buffer; la/lb[0]: k-half 0 in the OTHER buffer (next K-step)
    if constexpr (h == 0) {
        if constexpr (n < 16 && (n & 1)) AV_W_RD(FB[1][n >> 1], lb[1], WP_BOFF(n >> 1));
        if constexpr (n == 20) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (n == 21) __builtin_amdgcn_s_barrier();
        if constexpr (n >= 22 && n < 38 && !(n & 1)) AV_W_LD(vB[(n - 22) >> 1], pB, m0B + ((n - 22) >> 1) * 1024);
        if constexpr (n >= 23 && n < 39 && (n & 1)) AV_W_RD(FA[1][(n - 23) >> 1], la[1], ((n - 23) >> 1) * 2048);
        if constexpr (n == 50) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (n == 51) __builtin_amdgcn_s_barrier();
        if constexpr (n == 52 || n == 55 || n == 58 || n == 61) AV_W_LD(vA[(n - 52) / 3], pA, m0A + ((n - 52) / 3) * 1024);
    } else {
        if constexpr (n == 0) AV_W_LD(vA[4], pA, m0A + 4 * 1024);
        if constexpr (n == 26) { if constexpr (MODE == 2) asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); }
        if constexpr (n == 27) __builtin_amdgcn_s_barrier();
        if constexpr (n >= 28 && n < 36) AV_W_RD(FB[0][n - 28], lb[0], WP_BOFF(n - 28));
        if constexpr (n >= 37 && n < 53 && (n & 1)) AV_W_RD(FA[0][(n - 37) >> 1], la[0], ((n - 37) >> 1) * 2048);
        if constexpr (n == 32 || n == 40 || n == 48) AV_W_LD(vA[5 + (n - 32) / 8], pA, m0A + (5 + (n - 32) / 8) * 1024);
        if constexpr (n == 62) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int MODE, int... Ns>
__device__ __forceinline__ void wpgemm_kstep(std::integer_sequence<int, Ns...>, f32x4 (&acc)[8][8], bf16x8 (&FA)[2][8], bf16x8 (&FB)[2][8], const int (&la)[2],
                                             const int (&lb)[2], const unsigned (&vA)[8], const unsigned (&vB)[8], const bf16* pA, const bf16* pB, int m0A, int m0B) {
    (wpgemm_step<Ns, MODE>(acc, FA, FB, la, lb, vA, vB, pA, pB, m0A, m0B), ...);
}

#ifndef WP_RES_DEPTH
#define WP_RES_DEPTH 7
#endif
constexpr int WP_LDS = 2 * HSTAGE;                                  // two stage buffers; the epilogue does not touch LDS

// Diagnostic build only (-DAVLLM_GEMM_STAMPS, tools/gemm_stamps.sh): s_memtime stamps at the K-step boundaries of the persistent kernel, summed
// per wave into a buffer nothing else reads: [workgroup][wave][0: first K-step of a tile, 1: second, 2: the others, 3: epilogue (VALU + store
// issue), 4: fragment re-read, 5: tiles, 6: s_memtime span of the wave, 7: s_memrealtime span].  The shipped kernel executes no stamp.
#ifdef AVLLM_GEMM_STAMPS
__device__ unsigned long long g_wp_stamps[256 * 4 * 8];
#define WP_STAMP(slot) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                            __builtin_amdgcn_sched_barrier(0); st_acc[slot] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define WP_STAMP(slot) do { } while (0)
#endif

template <bool HAS2>
__global__ __launch_bounds__(256, 1) void gemm_bf16_wp_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = (g.e.M + HBM_ - 1) / HBM_, tiles_n = (g.e.N + HBN_ - 1) / HBN_;
    const int ntiles = tiles_m * tiles_n, G = gridDim.x;             // G <= ntiles (dispatcher)
    const int nt1 = g.K / BK, nt = nt1 + g.K2 / BK;                  // >= 2 (dispatcher)
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int pc0 = (fq ^ (fr & 7)) << 4, pc1 = ((4 + fq) ^ (fr & 7)) << 4;
    // weight image: 16-byte chunk index ^ key(row), key(row) = (row & 3) | ((row >> 3) & 1) << 2, which keeps the remapped fragment reads
    // (rows 8 (fr >> 2) + (fr & 3) + 4e + 32p) bank-conflict free; the activation image keeps chunk ^ (row & 7)
    const int keyB = (fr & 3) | (((fr >> 2) & 1) << 2);
    const int pcB0 = (fq ^ keyB) << 4, pcB1 = ((4 + fq) ^ keyB) << 4;
    const int rowA = lds0 + (wr * 128 + fr) * 128, rowB = lds0 + HBM_ * 128 + (wc * 128 + 8 * (fr >> 2) + (fr & 3)) * 128;
    int la[2] = {rowA + HSTAGE + pc0, rowA + pc1}, lb[2] = {rowB + HSTAGE + pcB0, rowB + pcB1};     // current buffer = 0
    int bo = 0;                                                      // byte offset of the current buffer
    const int mw = lds0 + wave * 8192;
    const unsigned ch = ((lane & 7) ^ (lane >> 3)) << 4, r0 = wave * 64 + (lane >> 3);
    // source-side swizzle of the weight rows: LDS row wave*64 + 8q + (lane >> 3) has key ((lane >> 3) & 3) | (q & 1) << 2
    const unsigned chB0 = ((lane & 7) ^ ((lane >> 3) & 3)) << 4, chB1 = ((lane & 7) ^ (((lane >> 3) & 3) | 4)) << 4;

    // Load context: the tile whose K-steps are being fetched (up to two K-steps ahead of the tile being multiplied).  Offsets as in the
    // one-shot kernel; the second K segment's (LoRA: one K-step per tile) are rebuilt in the K-steps that need them.
    unsigned vA1[8], vB1[8];
    const bf16 *tA1, *tB1;
    int lvid = blockIdx.x, lt = 0, lm0 = 0, ln0 = 0;
    // virtual id (blockIdx.x + k * G) -> tile id.  With a 256-workgroup grid every FULL round hands XCD x (= vid % 8: round-robin dispatch) the 32
    // consecutive ids [256 r + 32 x, +32) = exactly one 8 x 4 tile block (tile_coords); contiguous per-XCD ranges over the whole launch
    // (xcd_remap) start mid-block whenever ntiles / 8 is not a multiple of 32 (gate|up: 172).  The ragged last round keeps xcd_remap.
    // Measured: gate|up 581 -> 575 us, d(down) 272.7 -> 269, lm_head 785 -> 774.
    auto tile_id = [&](int vid) __attribute__((always_inline)) {
        if (G == 256 && tiles_m <= 32 && tiles_m % 8 == 0 && !(g.dbg & 12)) {      // only where tile_coords walks blocks; tall shapes (M = 394000) measured 2 % slower this way
            const int full = (ntiles >> 8) << 8;
            if (vid < full) return (vid & ~255) + ((vid & 7) << 5) + ((vid & 255) >> 3);
            return full + xcd_remap(vid - full, ntiles - full);
        }
        return xcd_remap(vid, ntiles);
    };
    auto set_ctx = [&](int vid) __attribute__((always_inline)) {
        int tm, tn;
        tile_coords(tile_id(vid), tiles_m, tiles_n, tm, tn, !(g.dbg & 4), g.gw);
        lm0 = tm * HBM_; ln0 = tn * HBN_;
        const unsigned la2 = (unsigned)g.lda * 2, lb2 = (unsigned)g.ldb * 2;
        unsigned ma = g.e.M - 1 - lm0, mb = g.e.N - 1 - ln0;
        if (g.dbg & 0x20000) { ma = ma < 63 ? ma : 63; mb = mb < 63 ? mb : 63; }      // ... only the tile's first 64 rows: TCP misses that hit in L2
        if (g.dbg & 0x10000) { ma = 0; mb = 0; }                     // experiment builds: every lane fetches row 0 (always a cache hit): the memory path out of the picture
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned r = r0 + q * 8;
            vA1[q] = __umul24(r < ma ? r : ma, la2) + ch; vB1[q] = __umul24(r < mb ? r : mb, lb2) + ((q & 1) ? chB1 : chB0);
        }
        tA1 = g.A + (long)lm0 * g.lda; tB1 = g.B + (long)ln0 * g.ldb;
        if (g.dbg & 0x40000) { tA1 = g.A; tB1 = g.B; }                 // experiment builds: every tile multiplies the first 256 rows of both operands (every line a TCP miss and, for short K, an L2 hit)
    };
    auto advance = [&]() __attribute__((always_inline)) {            // after K-step lt of the context tile has been issued
        if (++lt < nt) return;
        if (lvid + G < ntiles) { lvid += G; set_ctx(lvid); lt = 0; }
        else lt = nt - 1;                                            // nothing left: keep re-fetching the last K-step (never read); the loop's vmcnt arithmetic stays uniform
    };
    auto seg2_offsets = [&](unsigned (&va)[8], unsigned (&vb)[8], const bf16*& pa, const bf16*& pb) __attribute__((always_inline)) {
        const unsigned la2 = (unsigned)g.lda2 * 2, lb2 = (unsigned)g.ldb2 * 2, ma = g.e.M - 1 - lm0, mb = g.e.N - 1 - ln0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned r = r0 + q * 8;
            va[q] = __umul24(r < ma ? r : ma, la2) + ch; vb[q] = __umul24(r < mb ? r : mb, lb2) + ((q & 1) ? chB1 : chB0);
        }
        pa = g.A2 + (long)lm0 * g.lda2 + (long)(lt - nt1) * BK; pb = g.B2 + (long)ln0 * g.ldb2 + (long)(lt - nt1) * BK;
    };
    set_ctx(lvid);
    {   // experiment: de-phase the workgroups (all tiles take the same time, so the whole grid otherwise reaches its epilogue store burst at once)
        // measured (tools/gemm_epi_experiment.sh, clip qkv 394000x2304x768): 1448 us in lockstep, 1292 us with two phases (4 or 8: the same);
        // a one-round launch only pays the delay, hence >= 6 rounds.  dbg bits 4.. override the phase count (1 = off)
        const int phases = ((g.dbg >> 4) & 0xff) ? ((g.dbg >> 4) & 0xff) : (ntiles >= 6 * G ? 2 : 1);
        if (phases > 1) {
            const int ph = (blockIdx.x >> 3) % phases;
            const int n = ph * (nt * 2100 + 11000) / (phases * 6400);
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(100);
        }
    }
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
        unsigned va[8], vb[8];
        const bf16 *pa = tA1 + (long)lt * BK, *pb = tB1 + (long)lt * BK;
#pragma unroll
        for (int q = 0; q < 8; ++q) { va[q] = vA1[q]; vb[q] = vB1[q]; }
        if constexpr (HAS2) { if (lt >= nt1) seg2_offsets(va, vb, pa, pb); }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            AV_W_LD(va[q], pa, mw + s * HSTAGE + q * 1024);
            AV_W_LD(vb[q], pb, mw + s * HSTAGE + HBM_ * 128 + q * 1024);
        }
        advance();
    }
    f32x4 acc[8][8];
    bf16x8 FA[2][8], FB[2][8];
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        int a0 = rowA + pc0;
        const int b0 = rowB + pcB0;
#define AV_WP_RDB(J) AV_W_RD(FB[0][J], b0, WP_BOFF(J))
        AV_WP_RDB(0); AV_WP_RDB(1); AV_WP_RDB(2); AV_WP_RDB(3); AV_WP_RDB(4); AV_WP_RDB(5); AV_WP_RDB(6); AV_WP_RDB(7);
#pragma unroll
        for (int i = 0; i < 8; ++i) { AV_W_RD(FA[0][i], a0, 0); a0 += 2048; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    using Seq = std::make_integer_sequence<int, 128>;
    auto kstep = [&](auto modec) __attribute__((always_inline)) {
        constexpr int FIRST = decltype(modec)::value;                // MODE of wpgemm_step
        if constexpr (HAS2) {
            unsigned va[8], vb[8];
            const bf16 *pa = tA1 + (long)lt * BK, *pb = tB1 + (long)lt * BK;
#pragma unroll
            for (int q = 0; q < 8; ++q) { va[q] = vA1[q]; vb[q] = vB1[q]; }
            if (lt >= nt1) seg2_offsets(va, vb, pa, pb);
            wpgemm_kstep<FIRST>(Seq{}, acc, FA, FB, la, lb, va, vb, pa, pb, mw + bo, mw + bo + HBM_ * 128);
        } else {
            wpgemm_kstep<FIRST>(Seq{}, acc, FA, FB, la, lb, vA1, vB1, tA1 + (long)lt * BK, tB1 + (long)lt * BK, mw + bo, mw + bo + HBM_ * 128);
        }
        const int d = bo ? -HSTAGE : HSTAGE;
        la[1] += d; lb[1] += d; la[0] -= d; lb[0] -= d; bo += d;
        advance();
    };
    bool stores_in_flight = false;                                   // the previous tile's epilogue issued all of its 32 stores
#ifdef AVLLM_GEMM_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev, st_t0, st_r0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0), "=s"(st_r0) :: "memory");
    st_prev = st_t0;
#endif
    for (int vid = blockIdx.x; vid < ntiles; vid += G) {
        if (stores_in_flight) kstep(std::integral_constant<int, 2>{});
        else kstep(std::integral_constant<int, 1>{});
        WP_STAMP(0);
#ifdef AVLLM_GEMM_STAMPS
        if (nt > 1) { kstep(std::integral_constant<int, 0>{}); WP_STAMP(1); }
#pragma unroll 1
        for (int t = 2; t < nt; ++t) kstep(std::integral_constant<int, 0>{});
        WP_STAMP(2);
#else
#pragma unroll 1
        for (int t = 1; t < nt; ++t) kstep(std::integral_constant<int, 0>{});
#endif
        // result latency of the last MFMAs (invisible to the compiler's hazard recogniser): nops, and every accumulator named as an in/out
        // operand so that compiler-generated readers stay behind them
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#define AV_W_PIN8(I) asm volatile("" : "+a"(acc[I][0]), "+a"(acc[I][1]), "+a"(acc[I][2]), "+a"(acc[I][3]), "+a"(acc[I][4]), "+a"(acc[I][5]), "+a"(acc[I][6]), "+a"(acc[I][7]))
        AV_W_PIN8(0); AV_W_PIN8(1); AV_W_PIN8(2); AV_W_PIN8(3); AV_W_PIN8(4); AV_W_PIN8(5); AV_W_PIN8(6); AV_W_PIN8(7);
#undef AV_W_PIN8
        // Epilogue straight from the accumulators: lane (fr, fq) holds, for row block i and column pair p, the 8 consecutive output columns
        // 32p + 8fq .. +7 of row 16i + fr (acc[i][2p] = the first four, acc[i][2p+1] = the last four): bias / activation / residual on the
        // fp32 values, one rounding, one 16-byte store per pair; a store instruction covers 16 rows x 64 contiguous bytes.
        int tm, tn;
        tile_coords(tile_id(vid), tiles_m, tiles_n, tm, tn, !(g.dbg & 4), g.gw);
        const int m0 = tm * HBM_ + wr * 128 + fr, n = tn * HBN_ + wc * 128 + fq * 8;
        const bool full = (tm + 1) * HBM_ <= g.e.M && (tn + 1) * HBN_ <= g.e.N && !(g.dbg & 3);      // wave-uniform: every lane stores all 32 chunks
        stores_in_flight = full;                                     // dbg bit 1: experiment, strict wait
        // (the bias cannot ride in the accumulators' initial value: an MFMA's C and D operands share one register-file bit, and D is an AGPR)
        float b[4][8];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int c = 0; c < 8; ++c) b[p][c] = 0.f;
            if (g.e.bias && n + 32 * p < g.e.N) load_f<8>((const bf16*)g.e.bias + n + 32 * p, b[p]);
        }
        // this lane's first row; item (i, p) is 16 i rows down and 32 p columns right: uniform (scalar) multiples of the row strides
        bf16* const cp0 = (bf16*)g.e.C + (long)m0 * g.e.ldc + n;
        const bf16* const rp0 = g.e.R ? (const bf16*)g.e.R + (long)m0 * g.e.ldr + n : nullptr;
        const long ldc = g.e.ldc, ldr = g.e.R ? g.e.ldr : 0;
        // edge tiles (and activation + residual, which no model call makes): per-chunk bounds tests
        auto run = [&](auto actc) __attribute__((always_inline)) {
            constexpr int ACT = decltype(actc)::value;
#pragma clang loop unroll(full)
            for (int i = 0; i < 8; ++i) {
                const bool mrow = m0 + i * 16 < g.e.M;
#pragma clang loop unroll(full)
                for (int p = 0; p < 4; ++p) {
                    asm volatile("" : "+a"(acc[i][2 * p]), "+a"(acc[i][2 * p + 1]));
                    if (mrow && n + 32 * p < g.e.N && !(g.dbg & 1))
                        epilogue_fast8<ACT>(acc[i][2 * p], acc[i][2 * p + 1], b[p], g.e.bias != nullptr, rp0 ? rp0 + (i * 16) * ldr + 32 * p : nullptr,
                                            cp0 + (i * 16) * ldc + 32 * p);
                }
            }
        };
        // full tile, no residual: 8 accumulator reads, 8 bias adds (only in the instantiation with a bias), the activation, 4 packed converts and
        // one 16-byte store per chunk -- no bounds test, no select, no branch (the general form above spends ~40 instructions per chunk:
        // tools/gemm_stamps.py measured the epilogue at a quarter of a K = 768 tile)
        auto run_full = [&](auto actc, auto hasbc) __attribute__((always_inline)) {
            constexpr int ACT = decltype(actc)::value;
            constexpr bool HASB = decltype(hasbc)::value;
#ifdef AVLLM_EXPERIMENT_KNOBS
            u32x4 fold = {0u, 0u, 0u, 0u};
#endif
#pragma clang loop unroll(full)
            for (int i = 0; i < 8; ++i) {
                bf16* const cpi = cp0 + (long)(i * 16) * ldc;
#pragma clang loop unroll(full)
                for (int p = 0; p < 4; ++p) {
                    // the accumulators are "redefined" here for the compiler: without it every epilogue variant's 256 accumulator reads are hoisted
                    // above the variant branch as common subexpressions -- 256 live registers, ~130 spills around the epilogue
                    asm volatile("" : "+a"(acc[i][2 * p]), "+a"(acc[i][2 * p + 1]));
                    const f32x4 lo = acc[i][2 * p], hi = acc[i][2 * p + 1];
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if constexpr (HASB) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] += b[p][c];
                    }
                    if constexpr (ACT != AV_ACT_NONE) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = act_apply_fast(v[c], ACT);
                    }
#ifdef AVLLM_EXPERIMENT_KNOBS
                    {   // store cache-policy experiment (bits 20..22 of GEMM_DBG): 1 = non-temporal builtin, 2.. = explicit policy bits
                        const int sm = (g.dbg >> 20) & 7;
                        if (sm) {
                            bf16x8 t;
#pragma unroll
                            for (int c = 0; c < 8; ++c) t[c] = (bf16)v[c];
                            const u32x4 tv = __builtin_bit_cast(u32x4, t);
                            bf16* sp = cpi + 32 * p;
                            if (sm == 1) __builtin_nontemporal_store(t, (bf16x8*)sp);
                            else if (sm == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(sp), "v"(tv) : "memory");
                            else if (sm == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(sp), "v"(tv) : "memory");
                            else if (sm == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(sp), "v"(tv) : "memory");
                            else if (sm == 5) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(sp), "v"(tv) : "memory");
                            else if (sm == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(sp), "v"(tv) : "memory");
                            else {                                   // 7: all of the epilogue's arithmetic, ONE store per lane and tile (the folded chunks)
                                fold[0] ^= tv[0]; fold[1] ^= tv[1]; fold[2] ^= tv[2]; fold[3] ^= tv[3];
                                if (i == 7 && p == 3) *(u32x4*)sp = fold;
                            }
                            continue;
                        }
                    }
#endif
                    store_f<8>(cpi + 32 * p, v);
                }
            }
        };
        // full tile + residual (out-projection, fc2, o, down, gradient sums): ALL 32 residual chunks of the lane are requested in one burst into
        // the 128 registers the operand fragments occupy during the K loop (dead here: the next tile's are re-read after the epilogue), then
        // consumed in issue order.  The tile pays one memory latency instead of one per row block (tools/gemm_stamps.py: 12 us of a 29 us
        // out-projection tile with the one-row-block-ahead form).  A load may alias a later store (in-place residual): every load is issued
        // before the first store, and a lane only ever reads the bytes it writes itself.  Behind chunk c's load at its wait: 31 - c younger
        // loads and the c stores issued so far = 31 operations, whatever c: one constant vmcnt.  asm loads + explicit waits because
        // compiler-visible loads consumed later leave the wait-count pass with pending state that reaches the K loop as vmcnt(0) per K-step.
        auto run_res_full = [&](auto hasbc) __attribute__((always_inline)) {
            constexpr bool HASB = decltype(hasbc)::value;
            constexpr int RD = WP_RES_DEPTH;                         // residual row blocks requested ahead of the one being stored (7 = the whole tile at once)
            u32x4 rr[RD + 1][4];
            auto fetch = [&](int i) __attribute__((always_inline)) {
                const bf16* rp = rp0 + (long)(i * 16) * ldr;
                asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"
                             "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
                             : "=&v"(rr[i % (RD + 1)][0]), "=&v"(rr[i % (RD + 1)][1]), "=&v"(rr[i % (RD + 1)][2]), "=&v"(rr[i % (RD + 1)][3]) : "v"(rp) : "memory");
            };
#pragma clang loop unroll(full)
            for (int i = 0; i < RD && i < 8; ++i) fetch(i);
#pragma clang loop unroll(full)
            for (int i = 0; i < 8; ++i) {
                if (i + RD < 8) fetch(i + RD);
                bf16* const cpi = cp0 + (long)(i * 16) * ldc;
                // younger operations behind row block i's loads at their wait: 3 of its own asm statement, 4 per later row block in flight,
                // 4 stores per row block stored since the loads were issued (all of them = 31 with the whole tile in flight)
                constexpr int dummy = 0; (void)dummy;
                const int ahead = (i + RD < 8 ? i + RD : 7) - i, behind = RD < i ? RD : i;
#pragma clang loop unroll(full)
                for (int p = 0; p < 4; ++p) {
                    u32x4& x = rr[i % (RD + 1)][p];
                    const int nw = 3 + 4 * ahead + 4 * behind;
                    if (nw == 7) asm volatile("s_waitcnt vmcnt(7)" : "+v"(x) :: "memory");
                    else if (nw == 11) asm volatile("s_waitcnt vmcnt(11)" : "+v"(x) :: "memory");
                    else if (nw == 15) asm volatile("s_waitcnt vmcnt(15)" : "+v"(x) :: "memory");
                    else if (nw == 19) asm volatile("s_waitcnt vmcnt(19)" : "+v"(x) :: "memory");
                    else if (nw == 23) asm volatile("s_waitcnt vmcnt(23)" : "+v"(x) :: "memory");
                    else if (nw == 27) asm volatile("s_waitcnt vmcnt(27)" : "+v"(x) :: "memory");
                    else if (nw == 31) asm volatile("s_waitcnt vmcnt(31)" : "+v"(x) :: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(x) :: "memory");
                    const bf16x8 r = __builtin_bit_cast(bf16x8, x);
                    asm volatile("" : "+a"(acc[i][2 * p]), "+a"(acc[i][2 * p + 1]));
                    const f32x4 lo = acc[i][2 * p], hi = acc[i][2 * p + 1];
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if constexpr (HASB) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] += b[p][c];
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] += (float)r[c];
                    store_f<8>(cpi + 32 * p, v);
                }
            }
        };
        // the combinations the model calls make get the lean forms; anything else (and every edge tile) the general one
        const std::true_type yes{};
        const std::false_type no{};
        const bool hb = g.e.bias != nullptr;
#ifdef AVLLM_EXPERIMENT_KNOBS
        if (g.dbg & 0x80000) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // experiment: no operand load in flight while the epilogue stores
#endif
        if (full && !g.e.R && g.e.act == AV_ACT_NONE && !hb) run_full(std::integral_constant<int, AV_ACT_NONE>{}, no);           // Llama projections, gradient GEMMs
        else if (full && !g.e.R && g.e.act == AV_ACT_NONE && hb) run_full(std::integral_constant<int, AV_ACT_NONE>{}, yes);     // encoder q|k|v
        else if (full && !g.e.R && g.e.act == AV_ACT_GELU && hb) run_full(std::integral_constant<int, AV_ACT_GELU>{}, yes);     // Whisper fc1
        else if (full && !g.e.R && g.e.act == AV_ACT_QUICK_GELU && hb) run_full(std::integral_constant<int, AV_ACT_QUICK_GELU>{}, yes);      // CLIP fc1
        else if (full && g.e.R && g.e.act == AV_ACT_NONE && hb) run_res_full(yes);                                                // encoder out-projection / fc2
        else if (full && g.e.R && g.e.act == AV_ACT_NONE) run_res_full(no);                                                       // Llama o / down, gradient sums
        else if (g.e.act == AV_ACT_NONE) run(std::integral_constant<int, AV_ACT_NONE>{});
        else if (g.e.act == AV_ACT_GELU) run(std::integral_constant<int, AV_ACT_GELU>{});
        else if (g.e.act == AV_ACT_QUICK_GELU) run(std::integral_constant<int, AV_ACT_QUICK_GELU>{});
        else run(std::integral_constant<int, AV_ACT_SILU>{});
        WP_STAMP(3);
        // The next tile's k-half-0 fragments are read again here (its first K-step is complete in the current buffer: the last K-step's
        // barrier covered it) instead of being carried through the epilogue: 64 registers the epilogue code can use
        // (unconditional: after the workgroup's last tile the reads fetch stale LDS bytes nobody uses -- a conditional re-read keeps the 64
        // fragment registers live through the epilogue on the not-taken path, which the residual burst above needs)
        {
            int a0 = la[1] - pc1 + pc0;
            const int b0 = lb[1] - pcB1 + pcB0;
            AV_WP_RDB(0); AV_WP_RDB(1); AV_WP_RDB(2); AV_WP_RDB(3); AV_WP_RDB(4); AV_WP_RDB(5); AV_WP_RDB(6); AV_WP_RDB(7);
#pragma unroll
            for (int i = 0; i < 8; ++i) { AV_W_RD(FA[0][i], a0, 0); a0 += 2048; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        WP_STAMP(4);
#ifdef AVLLM_GEMM_STAMPS
        st_acc[5] += 1;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // in-flight DMA writes must not outlive the workgroup's LDS allocation
#ifdef AVLLM_GEMM_STAMPS
    {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
        if (lane == 0 && blockIdx.x < 256) {
            unsigned long long* o = g_wp_stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
            for (int i = 0; i < 6; ++i) o[i] = st_acc[i];
            o[6] = t1 - st_t0; o[7] = r1 - st_r0;
        }
    }
#endif
#undef AV_WP_RDB
#undef AV_W_RD
#undef AV_W_LD
}

// Persistent form of the 16-wave kernel (gemm_bf16_h_kernel): one workgroup per CU walks tiles id, id+G, id+2G, ... and issues the FIRST K-tile of its
// next output tile during the LAST K-step of the current one, so the per-tile prologue latency (exposed above, because a
// 128 KiB workgroup has no co-resident partner) hides under compute and the epilogue stores overlap the next tile's loads.
// Matters most for the short-K encoder GEMMs (K = 768: 12 K-steps per tile).
__global__ __launch_bounds__(1024, 4) void gemm_bf16_hp_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_m = (g.e.M + HBM_ - 1) / HBM_, tiles_n = (g.e.N + HBN_ - 1) / HBN_;
    const int ntiles = tiles_m * tiles_n, G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const int nt1 = g.K / BK, nt = nt1 + g.K2 / BK;
    const int fr = lane & 15, fq = lane >> 4;

    auto stage = [&](int m0, int n0, int t, int buf) {
        char* a_lds = smem + buf * HSTAGE;
        char* b_lds = a_lds + HBM_ * BK * 2;
        const bf16* Ap = t < nt1 ? g.A : g.A2;
        const bf16* Bp = t < nt1 ? g.B : g.B2;
        const long la = t < nt1 ? g.lda : g.lda2, lb = t < nt1 ? g.ldb : g.ldb2;
        const int k0 = (t < nt1 ? t : t - nt1) * BK;
#pragma unroll
        for (int p = 0; p < 2; ++p) stage_rows8(Ap, la, m0, g.e.M, k0, a_lds, p * 16 + wave, lane);
#pragma unroll
        for (int p = 0; p < 2; ++p) stage_rows8(Bp, lb, n0, g.e.N, k0, b_lds, p * 16 + wave, lane);
    };

    int tm, tn;
    tile_coords(first, tiles_m, tiles_n, tm, tn);
    int m0 = tm * HBM_, n0 = tn * HBN_;
    int cur = 0;
    stage(m0, n0, 0, 0);
    __syncthreads();
    for (int tile = first; tile < ntiles; tile += G) {
        const int next = tile + G;
        int nm0 = 0, nn0 = 0;
        if (next < ntiles) { int a, b; tile_coords(next, tiles_m, tiles_n, a, b); nm0 = a * HBM_; nn0 = b * HBN_; }
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nt; ++t) {
            if (t + 1 < nt) stage(m0, n0, t + 1, cur ^ 1);
            else if (next < ntiles) stage(nm0, nn0, 0, cur ^ 1);
            const char* a_lds = smem + cur * HSTAGE;
            const char* b_lds = a_lds + HBM_ * BK * 2;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 xa[4], wb[4];
                const int chunk = ks * 4 + fq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = wm * 64 + i * 16 + fr;
                    xa[i] = *(const bf16x8*)(a_lds + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = wn * 64 + j * 16 + fr;
                    wb[j] = *(const bf16x8*)(b_lds + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            cur ^= 1;
        }
#pragma clang loop unroll(full)
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + fr;
#pragma clang loop unroll(full)
            for (int j = 0; j < 4; ++j) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                epilogue_store4<bf16>(g.e, m, n0 + wn * 64 + j * 16 + fq * 4, v);
            }
        }
        m0 = nm0; n0 = nn0;
    }
}

// ------------------------------------------------------------------------------------------ bf16, N == 64 (LoRA rank side)
// C[M,64] = alpha * A[M,K] . B[64,K]^T.  A 128x128 tiling leaves 16 workgroups walking K serially (72 us measured at
// M=2048, K=4096).  Here a workgroup owns 16 rows and splits K over its 8 waves (split-K inside the block, reduced
// through LDS), operands go straight from global/L2 to MFMA fragments (no reuse to stage for: A is read once).
constexpr int SK_WAVES = 8;
__device__ __forceinline__ bf16x8 drop_frag(bf16x8 x, uint32_t seed, unsigned long long idx0, uint32_t thr, float sc) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
    av_mask8(v, seed, idx0, thr, sc);
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16)v[j];
    return r;
}

// DROP: the A operand is dropout(A) with the library's counter-based mask over the logical index row*K + col (A must be the
// full [M,K] activation) -- peft's lora_A(dropout(x)) without materialising dropout(x).
// NV = 16-column tiles that hold real rank columns (rank 16 padded to 64 -> NV = 1): only those weight rows are streamed and
// multiplied, the padding columns of the output are written as zeros.
template <bool DROP, int NV>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny64_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ B,
                                                                      long ldb, int M, int K, float alpha, void* __restrict__ C, long ldc,
                                                                      int out_f32, uint32_t seed_off, float p, const uint32_t* seed_dev) {
    __shared__ float part[SK_WAVES][16][NV * 16 + 1];
    const uint32_t seed = DROP ? av_seed(seed_dev, seed_off) : 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * 16;
    int ar = m0 + fr; ar = ar < M ? ar : M - 1;
    const int kw = K / SK_WAVES;                    // multiple of 32
    const bf16* ap = A + (long)ar * lda + (long)w * kw + fq * 8;
    const bf16* bp = B + (long)fr * ldb + (long)w * kw + fq * 8;
    f32x4 acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // explicit software unroll (the pragma form is rejected for the runtime trip count): 16-20 independent 16-byte loads are
    // issued before the first MFMA of a group consumes one, which is what hides the HBM/L2 latency here
    constexpr int UNR = NV == 1 ? 8 : 4;
    const uint32_t thr = av_drop_thr(p);
    const float dsc = av_drop_scale(p);
    const unsigned long long idx_base = (unsigned long long)ar * K + (unsigned long long)w * kw + fq * 8;
    int k = 0;
    for (; k + 32 * UNR <= kw; k += 32 * UNR) {
        bf16x8 xa[UNR], wb[UNR][NV];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            xa[u] = *(const bf16x8*)(ap + k + 32 * u);
            if (DROP) xa[u] = drop_frag(xa[u], seed, idx_base + k + 32 * u, thr, dsc);
#pragma unroll
            for (int j = 0; j < NV; ++j) wb[u][j] = *(const bf16x8*)(bp + (long)j * 16 * ldb + k + 32 * u);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int j = 0; j < NV; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[u][j], xa[u], acc[j], 0, 0, 0);     // D[n][m]
    }
    for (; k < kw; k += 32) {
        bf16x8 xa = *(const bf16x8*)(ap + k);
        if (DROP) xa = drop_frag(xa, seed, idx_base + k, thr, dsc);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bf16x8 wb = *(const bf16x8*)(bp + (long)j * 16 * ldb + k);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xa, acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[w][fr][j * 16 + fq * 4 + i] = acc[j][i];
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * 64; e += SK_WAVES * 64) {
        const int r = e >> 6, c = e & 63;
        if (m0 + r >= M) continue;
        float s = 0.f;
        if (c < NV * 16) {
#pragma unroll
            for (int x = 0; x < SK_WAVES; ++x) s += part[x][r][c];
            s *= alpha;
        }
        if (out_f32) ((float*)C)[(long)(m0 + r) * ldc + c] = s;
        else ((bf16*)C)[(long)(m0 + r) * ldc + c] = (bf16)s;
    }
}

// ------------------------------------------------------------------------------------------ bf16, M <= 16 (greedy decode steps)
// HBM-bound weight streaming: every decode step reads all 13.5 GB of bf16 Llama-2-7B weights once, whatever the batch
// (SURVEY.md §8d).  A workgroup owns 16 output columns (16 rows of W, streamed once, straight from global into MFMA
// fragments -- no LDS round trip for data without reuse), splits K over its 8 waves with 8 x 2 KiB of loads in flight per wave,
// and reduces the 8 partial 16x16 tiles through LDS.  The (<=16) activation rows ride along as the other MFMA operand.
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_smallm_kernel(GemmArgs g) {
    __shared__ float part[SK_WAVES][16][17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int M = g.e.M;
    const int ar = fr < M ? fr : M - 1;
    const int kw = g.K / SK_WAVES;
    const bf16* ap = g.A + (long)ar * g.lda + (long)w * kw + fq * 8;
    const bf16* bp = g.B + (long)(n0 + fr) * g.ldb + (long)w * kw + fq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 256 <= kw; k += 256) {          // explicit 8-deep unroll: 16 loads (16 KiB per wave) in flight
        bf16x8 wb[8], xa[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { wb[u] = *(const bf16x8*)(bp + k + 32 * u); xa[u] = *(const bf16x8*)(ap + k + 32 * u); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[u], xa[u], acc, 0, 0, 0);          // D[n][m]
    }
    for (; k < kw; k += 32)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(bp + k), *(const bf16x8*)(ap + k), acc, 0, 0, 0);
    if (w == 0 && g.K2 > 0) {
        const bf16* ap2 = g.A2 + (long)ar * g.lda2 + fq * 8;
        const bf16* bp2 = g.B2 + (long)(n0 + fr) * g.ldb2 + fq * 8;
        for (int k = 0; k < g.K2; k += 32)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(bp2 + k), *(const bf16x8*)(ap2 + k), acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) part[w][fq * 4 + i][fr] = acc[i];                      // [n][m]
    __syncthreads();
    if (threadIdx.x < 256) {
        const int m = threadIdx.x >> 4, nn = threadIdx.x & 15, n = n0 + nn;
        if (m < M && n < g.e.N) {
            float s = 0.f;
#pragma unroll
            for (int x = 0; x < SK_WAVES; ++x) s += part[x][nn][m];
            s = s * g.e.alpha + (g.e.bias ? to_f(((const bf16*)g.e.bias)[n]) : 0.f);
            s = act_apply(s, g.e.act);
            if (g.e.R) s += to_f(((const bf16*)g.e.R)[(long)(g.e.r_mod > 0 ? m % g.e.r_mod : m) * g.e.ldr + n]);
            if (g.e.out_f32) ((float*)g.e.C)[(long)m * g.e.ldc + n] = s;
            else ((bf16*)g.e.C)[(long)m * g.e.ldc + n] = (bf16)s;
        }
    }
}

// ------------------------------------------------------------------------------------------ f32
constexpr int FM = 64, FN = 64, FK = 16, FLD = FK + 1;

struct GemmArgsF {
    const float* A; const float* B; const float* A2; const float* B2;
    long lda, ldb, lda2, ldb2;
    int K, K2;
    EpiParams e;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgsF g) {
    __shared__ float As[FM * FLD], Bs[FN * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (g.e.M + FM - 1) / FM, tiles_n = (g.e.N + FN - 1) / FN;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * FM, n0 = tn * FN;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nt1 = g.K / FK, nt = nt1 + g.K2 / FK;
    const int lr = tid >> 2, lc = (tid & 3) * 4;       // 64 rows x 4 float4 per operand tile
    const int fr = lane & 15, fq = lane >> 4;
    for (int t = 0; t < nt; ++t) {
        const float* Ap; const float* Bp; long la, lb; int k0;
        if (t < nt1) { Ap = g.A; Bp = g.B; la = g.lda; lb = g.ldb; k0 = t * FK; }
        else { Ap = g.A2; Bp = g.B2; la = g.lda2; lb = g.ldb2; k0 = (t - nt1) * FK; }
        int ar = m0 + lr; ar = ar < g.e.M ? ar : g.e.M - 1;
        int br = n0 + lr; br = br < g.e.N ? br : g.e.N - 1;
        const f32x4 av = *(const f32x4*)(Ap + (long)ar * la + k0 + lc);
        const f32x4 bv = *(const f32x4*)(Bp + (long)br * lb + k0 + lc);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) { As[lr * FLD + lc + i] = av[i]; Bs[lr * FLD + lc + i] = bv[i]; }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < FK / 4; ++ks) {
            float xa[2], wb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) xa[i] = As[(wm * 32 + i * 16 + fr) * FLD + ks * 4 + fq];
#pragma unroll
            for (int j = 0; j < 2; ++j) wb[j] = Bs[(wn * 32 + j * 16 + fr) * FLD + ks * 4 + fq];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            epilogue_store4<float>(g.e, m, n0 + wn * 32 + j * 16 + fq * 4, v);
        }
    }
}

}  // namespace

// Per-device launch state: the CU count and the "dynamic LDS limit raised" flags belong to the device the call runs on
// (one process may drive several GPUs; hipFuncSetAttribute applies to the current device's copy of the code object).
struct GemmDevState { int ncu = 0; bool attr_base = false, attr_wp = false, attr_w = false, attr_h = false; };
static GemmDevState g_gemm_dev[64];
static GemmDevState* gemm_dev_state() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    GemmDevState* s = &g_gemm_dev[dev & 63];
    if (!s->ncu && hipDeviceGetAttribute(&s->ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) s->ncu = 256;
    return s;
}
#ifdef AVLLM_GEMM_STAMPS
extern "C" int avllm_debug_read_gemm_stamps(unsigned long long* host, int32_t n) {
    AV_HIP(hipDeviceSynchronize());
    AV_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wp_stamps), sizeof(unsigned long long) * (size_t)(n < 256 * 4 * 8 ? n : 256 * 4 * 8)));
    return AV_OK;
}
#endif
static int g_gemm_variant = -1;     // 0 auto, 1 128x128, 2 256x128 ring, 5 256x256 16-wave, 6 256x256 16-wave persistent, 7 256x256 4-wave, 8 256x256 4-wave persistent (lean epilogue only), 9 256x128 persistent, two workgroups per CU (gemm_dp.hip; lean epilogue only)
bool av_gemm_dp_ok(const avllm_gemm_desc* d);
int av_gemm_dp(const avllm_gemm_desc* d, hipStream_t st, int dbg);
extern "C" int avllm_set_gemm_variant(int v) { g_gemm_variant = v; return 0; }
bool av_prof_enabled();
void av_prof_before(hipStream_t st);
void av_prof_after(hipStream_t st, double flops);

int av_gemm(const avllm_gemm_desc* d, hipStream_t st) {
    AV_CHECK_ARG(d && d->A && d->B && d->C, "gemm: null operand");
    AV_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    AV_CHECK_ARG(d->K % 64 == 0 && d->K2 % 64 == 0, "gemm: K (%d) and K2 (%d) must be multiples of 64", d->K, d->K2);
    AV_CHECK_ARG(d->K2 == 0 || (d->A2 && d->B2), "gemm: K2>0 needs A2/B2");
    AV_CHECK_ARG(d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 4 == 0, "gemm: leading dims must keep 16-byte rows");
    AV_CHECK_ARG(d->dtype == AV_F32 || d->dtype == AV_BF16, "gemm: dtype %d", d->dtype);
    EpiParams e;
    e.C = d->C; e.ldc = d->ldc; e.out_f32 = d->out_f32 || d->dtype == AV_F32; e.bias = d->bias; e.R = d->R; e.ldr = d->ldr;
    e.r_mod = d->r_mod; e.g_in = d->g_in; e.g_out = d->g_out; e.g_off = d->g_off;
    e.alpha = d->alpha; e.act = d->act; e.M = d->M; e.N = d->N;
    e.drop_seed = d->drop_seed; e.drop_p = d->drop_p; e.seed_dev = d->seed_dev;
    const bool prof = av_prof_enabled();
    if (prof) av_prof_before(st);
    const size_t osz = e.out_f32 ? 4 : 2;
    const bool wide_ok = d->dtype == AV_BF16 && d->N % 8 == 0 && ((uintptr_t)d->C % 16 == 0) && (d->ldc * osz) % 16 == 0 &&
                         (!d->bias || (uintptr_t)d->bias % 16 == 0) && (!d->R || ((uintptr_t)d->R % 16 == 0 && d->ldr % 8 == 0)) &&
                         !av_knob(AV_KNOB_NARROW_EPILOGUE);
    if (d->dtype == AV_BF16 && d->M <= 16 && d->N % 16 == 0 && d->K % (32 * SK_WAVES) == 0 && d->K2 % 32 == 0 && d->g_in == 0 &&
        d->drop_p <= 0.f && d->a_drop_p <= 0.f && g_gemm_variant <= 0) {      // a_drop: the rank-side kernel below owns the fused mask
        GemmArgs g;
        g.A = (const bf16*)d->A; g.B = (const bf16*)d->B; g.A2 = (const bf16*)d->A2; g.B2 = (const bf16*)d->B2;
        g.lda = d->lda; g.ldb = d->ldb; g.lda2 = d->lda2; g.ldb2 = d->ldb2; g.K = d->K; g.K2 = d->K2; g.e = e;
        g.wide_epi = wide_ok; g.dbg = 0; g.gw = 0;
        hipLaunchKernelGGL(gemm_smallm_kernel, dim3(d->N / 16), dim3(SK_WAVES * 64), 0, st, g);
    } else if (d->dtype == AV_BF16 && d->N == 64 && d->K2 == 0 && !d->bias && !d->R && d->act == AV_ACT_NONE && d->g_in == 0 && d->drop_p <= 0.f &&
               d->K % (32 * SK_WAVES) == 0 && (d->M >= 256 || d->a_drop_p > 0.f)) {
        AV_CHECK_ARG(d->a_drop_p <= 0.f || d->lda == d->K, "gemm: a_drop needs the full contiguous [M,K] activation as A");
        AV_CHECK_ARG(d->n_valid >= 0 && d->n_valid <= 64, "gemm: n_valid=%d", d->n_valid);
        const int nv = d->n_valid > 0 ? (d->n_valid + 15) / 16 : 4;
#define AV_SKINNY(DROPV, NVV, SEED, P) hipLaunchKernelGGL((gemm_skinny64_kernel<DROPV, NVV>), dim3(av_cdiv(d->M, 16)), dim3(SK_WAVES * 64), 0, st, \
            (const bf16*)d->A, d->lda, (const bf16*)d->B, d->ldb, d->M, d->K, d->alpha, d->C, d->ldc, e.out_f32, SEED, P, d->seed_dev)
        if (d->a_drop_p > 0.f) {
            if (nv == 1) AV_SKINNY(true, 1, d->a_drop_seed, d->a_drop_p);
            else if (nv == 2) AV_SKINNY(true, 2, d->a_drop_seed, d->a_drop_p);
            else AV_SKINNY(true, 4, d->a_drop_seed, d->a_drop_p);
        } else {
            if (nv == 1) AV_SKINNY(false, 1, 0u, 0.f);
            else if (nv == 2) AV_SKINNY(false, 2, 0u, 0.f);
            else AV_SKINNY(false, 4, 0u, 0.f);
        }
#undef AV_SKINNY
    } else if (d->a_drop_p > 0.f) {
        return av_set_error(AV_ERR_UNSUPPORTED, "gemm: a_drop_p is only implemented by the bf16 N==64 rank-side kernel (K %% 256 == 0)");
    } else if (d->dtype == AV_BF16) {
        GemmArgs g;
        g.A = (const bf16*)d->A; g.B = (const bf16*)d->B; g.A2 = (const bf16*)d->A2; g.B2 = (const bf16*)d->B2;
        g.lda = d->lda; g.ldb = d->ldb; g.lda2 = d->lda2; g.ldb2 = d->ldb2; g.K = d->K; g.K2 = d->K2; g.e = e;
        g.wide_epi = wide_ok;
        {   // Tall shapes can walk the tiles in column groups whose weight panels fit an XCD's L2 beside the streaming activation panels.
            // Measured (profiles/r03_pmc_gemm_shapes.txt, gpurun_out/r3_gw_sweep.log): the fabric reads of the CLIP fc1 fall from 8.8x to 4.6x
            // the algorithmic bytes (qkv 5.3x -> 4.1x) and the launch gets no faster (fc1 1688 -> 1670 us at width 6, 1751 at 4; qkv 1263 ->
            // 1293 .. 1324): these launches are not bound by their read traffic.  So the walk stays row-major; knob GEMM_GW = n forces a width.
            const int forced = av_knob(AV_KNOB_GEMM_GW);
            g.gw = forced > 0 ? forced : 0;
        }
#ifdef AVLLM_EXPERIMENT_KNOBS
        g.dbg = av_knob(AV_KNOB_GEMM_DBG);
#else
        g.dbg = 0;
#endif
        GemmDevState* ds = gemm_dev_state();
        if (!ds->attr_base) {
            AV_HIP(hipFuncSetAttribute((const void*)gemm_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES));
            AV_HIP(hipFuncSetAttribute((const void*)gemm_bf16_l_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LNSTAGE * LSTAGE));
            ds->attr_base = true;
        }
        const int variant = g_gemm_variant >= 0 ? g_gemm_variant : (g_gemm_variant = getenv("AVLLM_GEMM_VARIANT") ? atoi(getenv("AVLLM_GEMM_VARIANT")) : 0);
        constexpr int XBM = 256, XBN = 256;
        const int xtiles = av_cdiv(d->M, XBM) * av_cdiv(d->N, XBN);
        // automatic choice (tools/gemm_bench.py on MI355X): 256x256 / 16 waves whenever it fills the chip, 256x128 ring for very
        // long K with few tiles, else 128x128 with two workgroups per CU
        // the 16-wave kernel keeps 32-bit element offsets of its staging rows: operands must stay below 2^32 elements
        const bool fits32 = (double)d->M * (double)(d->lda > d->lda2 ? d->lda : d->lda2) < 4.0e9 &&
                            (double)d->N * (double)(d->ldb > d->ldb2 ? d->ldb : d->ldb2) < 4.0e9;
        const bool auto_h = variant == 0 && xtiles >= 200 && fits32;   // 4-wave kernel for long K (fixed cost 10.9 us per tile + 1.43 us per K-step against
        // 7.7 + 1.67 for the 16-wave kernel, tools/gemm_ktile_sweep.py; small grids favour the 16-wave kernel a little longer)
        const bool auto_l = variant == 0 && !auto_h && d->K >= 16384;
        const bool lean_ok = wide_ok && d->alpha == 1.f && d->drop_p <= 0.f && d->g_in <= 0 && !e.out_f32 && d->r_mod <= 0;
        if (d->M > 128 && variant == 9 && lean_ok && av_gemm_dp_ok(d)) {           // A/B variant, never chosen automatically (gemm_dp.hip: why)
            AV_TRY(av_gemm_dp(d, st, g.dbg));
        } else if (d->M > 128 && (variant == 8 || auto_h) && lean_ok && d->K + d->K2 >= 128) {
            if (!ds->attr_wp) {
                AV_HIP(hipFuncSetAttribute((const void*)gemm_bf16_wp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS));
                AV_HIP(hipFuncSetAttribute((const void*)gemm_bf16_wp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS));
                ds->attr_wp = true;
            }
            const int ncu8 = ds->ncu;
            if (d->K2 > 0) hipLaunchKernelGGL(gemm_bf16_wp_kernel<true>, dim3(xtiles < ncu8 ? xtiles : ncu8), dim3(256), WP_LDS, st, g);
            else hipLaunchKernelGGL(gemm_bf16_wp_kernel<false>, dim3(xtiles < ncu8 ? xtiles : ncu8), dim3(256), WP_LDS, st, g);
        } else if (d->M > 128 && (variant == 7 || (auto_h && (d->K + d->K2 >= 4096 || (d->K + d->K2 >= 2048 && xtiles >= 1024)))) && d->K + d->K2 >= 128) {
            if (!ds->attr_w) { AV_HIP(hipFuncSetAttribute((const void*)gemm_bf16_w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HSTAGE)); ds->attr_w = true; }
            hipLaunchKernelGGL(gemm_bf16_w_kernel, dim3(xtiles), dim3(256), 2 * HSTAGE, st, g);
        } else if (d->M > 128 && (variant == 5 || variant == 6 || auto_h)) {
            AV_CHECK_ARG(fits32 || variant == 6, "gemm: operand too large for the 256x256 kernel's 32-bit row offsets");
            if (!ds->attr_h) {
                AV_HIP(hipFuncSetAttribute((const void*)gemm_bf16_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HSTAGE));
                AV_HIP(hipFuncSetAttribute((const void*)gemm_bf16_hp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HSTAGE));
                ds->attr_h = true;
            }
            const int ncu = ds->ncu;
            if (variant != 6) hipLaunchKernelGGL(gemm_bf16_h_kernel, dim3(xtiles), dim3(1024), 2 * HSTAGE, st, g);
            else hipLaunchKernelGGL(gemm_bf16_hp_kernel, dim3(xtiles < ncu ? xtiles : ncu), dim3(1024), 2 * HSTAGE, st, g);
        } else if (d->M > 128 && (variant == 2 || auto_l)) {
            const int tiles = av_cdiv(d->M, LBM) * av_cdiv(d->N, LBN);
            hipLaunchKernelGGL(gemm_bf16_l_kernel, dim3(tiles), dim3(512), LNSTAGE * LSTAGE, st, g);
        } else {
            const int tiles = av_cdiv(d->M, BM) * av_cdiv(d->N, BN);
            hipLaunchKernelGGL(gemm_bf16_kernel, dim3(tiles), dim3(256), 4 * TILE_BYTES, st, g);
        }
    } else {
        GemmArgsF g;
        g.A = (const float*)d->A; g.B = (const float*)d->B; g.A2 = (const float*)d->A2; g.B2 = (const float*)d->B2;
        g.lda = d->lda; g.ldb = d->ldb; g.lda2 = d->lda2; g.ldb2 = d->ldb2; g.K = d->K; g.K2 = d->K2; g.e = e;
        const int tiles = av_cdiv(d->M, FM) * av_cdiv(d->N, FN);
        hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, st, g);
    }
    if (prof) av_prof_after(st, 2.0 * d->M * (double)d->N * (double)(d->K + d->K2));
    AV_LAUNCH_CHECK();
    return AV_OK;
}
