// Flash attention backward for gfx950 (bf16, head_dim 128 and 64): recompute P from Q,K and the forward's LSE,
// no N x N score matrix in memory.  Autograd of HF's softmax(QK^T*scale + causal)V for the Llama blocks
// (reference: loss.backward(), trainer/clip_whisper_trainer.py:454).  Two deterministic kernels, no atomics:
//
//   dQ kernel   : wave = 32 queries (query on the MFMA lane, as in the forward).  S^T = K.Q^T and dP^T = V.dO^T
//                 leave P / dS with lane-local LSE and delta; dS^T registers are the B operand of
//                 dQ^T += K^T.dS^T, K^T fragments come from ds_read_b64_tr_b16.
//   dK/dV kernel: wave = 32 keys (key on the lane).  S = Q.K^T and dP = dO.V^T put the key on the lane, so P and
//                 dS registers are directly the B operands of dV^T += dO^T.P and dK^T += Q^T.dS
//                 (cdna guide App. B "Attention backward": key on the lane); Q^T / dO^T by transposed LDS reads.
//
// LDS: every tile is kept as a row image (row stride HD*2+16 B, conflict-free ds_read_b128) and, where a
// transposed operand is needed, as a second image with row stride HD*2+64 B (conflict-free 4-row tr reads).
#include "common.h"
#include "avllm_internal.h"

namespace {

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;
typedef __attribute__((ext_vector_type(8))) short short8v;

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// A operand = X^T fragment for the 32x32x16 MFMA: rows = 32 columns of the LDS tile starting at col0, k = 16 tile rows
// starting at row0 (permuted order matching an accumulator-as-B operand: element j <-> row 8(j>>2)+4half+(j&3))
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int stride, int row0, int col0, int lane) {
    const int g = lane >> 4, i16 = lane & 15, half = lane >> 5;
    const char* a0 = img + (row0 + 4 * half + (i16 >> 2)) * stride + (col0 + 16 * (g & 1) + 4 * (i16 & 3)) * 2;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0 + 8 * stride));
    const short8v both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

// Tiles go global -> registers -> LDS: the loads of the NEXT tile are issued right after the current tile has been written to
// LDS, so their latency runs under the current tile's MFMA work.  Both kernels are held to 256 registers (two waves per SIMD, i.e.
// two workgroups per CU): left alone hipcc takes ~400 and a lone wave per SIMD exposes every barrier and LDS round trip.
template <int HD, int NT, int ROWS> struct RowRegs { u32x4 v[(ROWS * (HD / 8) + NT - 1) / NT]; };

template <int HD, int NT, int ROWS>
__device__ __forceinline__ void load_rows(RowRegs<HD, NT, ROWS>& r, const bf16* __restrict__ src, long ld, int row0, int rows_max, int tid) {
    constexpr int CPR = HD / 8, N = (ROWS * CPR + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = tid + i * NT, row = c / CPR, ch = c % CPR;
        r.v[i] = (u32x4){0u, 0u, 0u, 0u};
        if (c < ROWS * CPR && row0 + row < rows_max) r.v[i] = *(const u32x4*)(src + (long)(row0 + row) * ld + ch * 8);
    }
}
template <int HD, int NT, int ROWS>
__device__ __forceinline__ void store_rows(const RowRegs<HD, NT, ROWS>& r, char* img_a, int stride_a, char* img_b, int stride_b, int tid) {
    constexpr int CPR = HD / 8, N = (ROWS * CPR + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = tid + i * NT, row = c / CPR, ch = c % CPR;
        if (c < ROWS * CPR) {
            if (img_a) *(u32x4*)(img_a + row * stride_a + ch * 16) = r.v[i];
            if (img_b) *(u32x4*)(img_b + row * stride_b + ch * 16) = r.v[i];
        }
    }
}

// Inverse rotary embedding on a gradient row held in the MFMA output layout (lane = row, registers [d][4*g4+i] = element
// 32d + 8g4 + 4half + i): element e < HD/2 pairs with e + HD/2, which the same lane holds in block d + HD/64.  table row t =
// (cos, sin) interleaved per frequency (elementwise.hip rope_table_kernel).  Saves the separate pass over dq|dk after the kernel.
template <int HD>
__device__ __forceinline__ void unrope(f32x16 (&acc)[HD / 32], const float* __restrict__ tab, int t, int half) {
    constexpr int HB = HD / 64;                                  // 32-wide blocks per half
    const float* row = tab + (long)t * (HD / 2) * 2;
#pragma unroll
    for (int d = 0; d < HB; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int e = 32 * d + 8 * g4 + 4 * half;
            const f32x4 c0 = *(const f32x4*)(row + 2 * e), c1 = *(const f32x4*)(row + 2 * e + 4);
            const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = acc[d][4 * g4 + i], b = acc[d + HB][4 * g4 + i], c = cs[2 * i], sn = cs[2 * i + 1];
                acc[d][4 * g4 + i] = a * c + b * sn;
                acc[d + HB][4 * g4 + i] = b * c - a * sn;
            }
        }
}

// ------------------------------------------------------------------------------------------------ dQ
template <int HD, int NW, bool CAUSAL>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq_mfma(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                            const bf16* __restrict__ v, const bf16* __restrict__ dout,
                                                            const float* __restrict__ lse, float* __restrict__ delta,
                                                            bf16* __restrict__ dq, int T, int H, long ldq, long ldk, long ldv,
                                                            long lddo, long lddq, float scale, int G, const bf16* __restrict__ o, long ldo,
                                                            const float* __restrict__ rope_tab) {
    constexpr int RS = HD * 2 + 16, TS = HD * 2 + 64, NT = NW * 64;
    __shared__ __attribute__((aligned(16))) char k_row[64 * RS];
    __shared__ __attribute__((aligned(16))) char k_tr[64 * TS];
    __shared__ __attribute__((aligned(16))) char v_row[64 * RS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int hh = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + w) * 32;
    const int qpos = q0 + r;
    const int qrow = qpos < T ? qpos : T - 1;
    const float sl = scale * 1.4426950408889634f;

    bf16x8 qf[HD / 16], dof[HD / 16];
    {
        const bf16* qp = q + ((long)b * T + qrow) * ldq + (long)hh * HD + 8 * half;
        const bf16* dp = dout + ((long)b * T + qrow) * lddo + (long)hh * HD + 8 * half;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) { qf[ks] = *(const bf16x8*)(qp + 16 * ks); dof[ks] = *(const bf16x8*)(dp + 16 * ks); }
    }
    const float lse2 = lse[((long)b * H + hh) * T + qrow] * 1.4426950408889634f;
    // delta = rowsum(dO * O): this kernel already holds its query's dO row (half of it per lane), so it computes delta itself and
    // publishes it for the dK/dV kernel that follows on the stream (one launch and one pass over O and dO fewer per layer)
    float dlt = 0.f;
    {
        const bf16* orow = o + ((long)b * T + qrow) * ldo + (long)hh * HD + 8 * half;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            const bf16x8 of = *(const bf16x8*)(orow + 16 * ks);
#pragma unroll
            for (int j = 0; j < 8; ++j) dlt += (float)dof[ks][j] * (float)of[j];
        }
        dlt += __shfl_xor(dlt, 32);
        if (half == 0 && qpos < T) delta[((long)b * H + hh) * T + qpos] = dlt;
    }
    f32x16 acc[HD / 32];
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[d][i] = 0.f;

    const int blk_qmax = min(T - 1, (int)(blockIdx.x * NW + NW) * 32 - 1);
    const int k_end = CAUSAL ? min(T, blk_qmax + 1) : T;
    const bool wave_active = q0 < T;
    const int wave_kmax = CAUSAL ? min(T - 1, q0 + 31) : T - 1;
    const bf16* kbase = k + (long)b * T * ldk + (long)(hh / G) * HD;      // grouped-query attention: G query heads per K/V head
    const bf16* vbase = v + (long)b * T * ldv + (long)(hh / G) * HD;

    RowRegs<HD, NT, 64> kr, vr;
    load_rows<HD, NT, 64>(kr, kbase, ldk, 0, T, tid);
    load_rows<HD, NT, 64>(vr, vbase, ldv, 0, T, tid);
    for (int kb = 0; kb < k_end; kb += 64) {
        __syncthreads();
        store_rows<HD, NT, 64>(kr, k_row, RS, k_tr, TS, tid);
        store_rows<HD, NT, 64>(vr, v_row, RS, nullptr, 0, tid);
        __syncthreads();
        if (kb + 64 < k_end) {
            load_rows<HD, NT, 64>(kr, kbase, ldk, kb + 64, T, tid);
            load_rows<HD, NT, 64>(vr, vbase, ldv, kb + 64, T, tid);
        }
        if (!wave_active || kb > wave_kmax) continue;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            if (kb + 32 * sb > wave_kmax || kb + 32 * sb >= T) break;
            f32x16 s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(k_row + (32 * sb + r) * RS + (2 * ks + half) * 16);
                const bf16x8 vf = *(const bf16x8*)(v_row + (32 * sb + r) * RS + (2 * ks + half) * 16);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp, 0, 0, 0);
            }
            const int klim = CAUSAL ? min(T - 1, qpos) : T - 1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kb + 32 * sb + acc_row(i, half);
                const float p = key <= klim ? __builtin_amdgcn_exp2f(s[i] * sl - lse2) : 0.f;
                s[i] = p * (dp[i] - dlt) * scale;                 // dS^T
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 dsb;
#pragma unroll
                for (int j = 0; j < 8; ++j) dsb[j] = (bf16)s[8 * st + j];
#pragma unroll
                for (int d = 0; d < HD / 32; ++d) {
                    const bf16x8 kt = tr_frag(k_tr, TS, 32 * sb + 16 * st, 32 * d, lane);
                    acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, dsb, acc[d], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_active || qpos >= T) return;
    if (rope_tab) unrope<HD>(acc, rope_tab, qpos, half);          // gradient w.r.t. the pre-RoPE q (transpose of the rotation)
    bf16* op = dq + ((long)b * T + qpos) * lddq + (long)hh * HD;
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float vals[4] = {acc[d][4 * g4], acc[d][4 * g4 + 1], acc[d][4 * g4 + 2], acc[d][4 * g4 + 3]};
            store_f<4>(op + 32 * d + 8 * g4 + 4 * half, vals);
        }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int HD, int NW, bool CAUSAL>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dkv_mfma(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                             const bf16* __restrict__ v, const bf16* __restrict__ dout,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             bf16* __restrict__ dk, bf16* __restrict__ dv, int T, int H, long ldq,
                                                             long ldk, long ldv, long lddo, long lddk, long lddv, float scale, int G,
                                                             const float* __restrict__ rope_tab) {
    constexpr int RS = HD * 2 + 16, TS = HD * 2 + 64, NT = NW * 64;
    __shared__ __attribute__((aligned(16))) char q_row[32 * RS];
    __shared__ __attribute__((aligned(16))) char q_tr[32 * TS];
    __shared__ __attribute__((aligned(16))) char do_row[32 * RS];
    __shared__ __attribute__((aligned(16))) char do_tr[32 * TS];
    __shared__ __attribute__((aligned(16))) float lse_s[32];
    __shared__ __attribute__((aligned(16))) float dlt_s[32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int hh = blockIdx.y, b = blockIdx.z;
    const int k0 = (blockIdx.x * NW + w) * 32;          // this wave's 32 keys
    const int kpos = k0 + r;
    const int krow = kpos < T ? kpos : T - 1;
    const float sl = scale * 1.4426950408889634f;

    // B operands: lane holds K[key r][16ks + 8half ..] in registers for the whole kernel.  The V fragments of the same shape are parked in a
    // wave-private LDS slice and read back where dP needs them (HD = 128: with both sets in registers the kernel needs ~282 VGPRs against the 256
    // of two waves per SIMD -- 26 spilled registers whose scratch traffic sat in the loop; rocprof: 62 % of the wave cycles waiting)
    constexpr bool V_LDS = HD > 64;
    __shared__ __attribute__((aligned(16))) char v_keep[V_LDS ? NW * 32 * RS : 16];
    char* const vk = v_keep + (V_LDS ? (w * 32 + r) * RS + half * 16 : 0);
    bf16x8 kf[HD / 16], vf[V_LDS ? 1 : HD / 16];
    {
        const bf16* kp = k + ((long)b * T + krow) * ldk + (long)hh * HD + 8 * half;
        const bf16* vp = v + ((long)b * T + krow) * ldv + (long)hh * HD + 8 * half;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            kf[ks] = *(const bf16x8*)(kp + 16 * ks);
            if constexpr (V_LDS) *(bf16x8*)(vk + 32 * ks) = *(const bf16x8*)(vp + 16 * ks);      // same lane reads it back: no barrier
            else vf[ks] = *(const bf16x8*)(vp + 16 * ks);
        }
    }
    f32x16 dka[HD / 32], dva[HD / 32];
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dka[d][i] = 0.f; dva[d][i] = 0.f; }

    const int blk_k0 = blockIdx.x * NW * 32;
    const int q_begin = CAUSAL ? (blk_k0 / 32) * 32 : 0;
    const bool wave_active = k0 < T;
    // hh is the K/V head (grid.y = H / G); the G query heads of its group are walked one after the other: iteration `it` is
    // query tile (it % ntile) of query head hh*G + it / ntile
    const int ntile = q_begin < T ? (T - q_begin + 31) / 32 : 0, total = ntile * G;

    RowRegs<HD, NT, 32> qr, dr;
    float lreg = 0.f, dreg = 0.f;
    auto fetch = [&](int it) {
        const int hq = hh * G + it / ntile, qb0 = q_begin + (it % ntile) * 32;
        load_rows<HD, NT, 32>(qr, q + (long)b * T * ldq + (long)hq * HD, ldq, qb0, T, tid);
        load_rows<HD, NT, 32>(dr, dout + (long)b * T * lddo + (long)hq * HD, lddo, qb0, T, tid);
        if (tid < 32) {
            const int qi = qb0 + tid;
            lreg = qi < T ? lse[((long)b * H + hq) * T + qi] * 1.4426950408889634f : 0.f;
            dreg = qi < T ? delta[((long)b * H + hq) * T + qi] : 0.f;
        }
    };
    if (total > 0) fetch(0);
    for (int it = 0; it < total; ++it) {
        const int qb = q_begin + (it % ntile) * 32;
        __syncthreads();
        store_rows<HD, NT, 32>(qr, q_row, RS, q_tr, TS, tid);
        store_rows<HD, NT, 32>(dr, do_row, RS, do_tr, TS, tid);
        if (tid < 32) { lse_s[tid] = lreg; dlt_s[tid] = dreg; }
        __syncthreads();
        if (it + 1 < total) fetch(it + 1);
        if (!wave_active || (CAUSAL && qb + 31 < k0)) continue;       // every query of this tile precedes the wave's keys
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            const bf16x8 qa = *(const bf16x8*)(q_row + r * RS + (2 * ks + half) * 16);
            const bf16x8 da = *(const bf16x8*)(do_row + r * RS + (2 * ks + half) * 16);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);       // S[query][key]
            bf16x8 vb;
            if constexpr (V_LDS) vb = *(const bf16x8*)(vk + 32 * ks); else vb = vf[ks];
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vb, dp, 0, 0, 0);         // dP[query][key]
        }
        // P and dS leave the fp32 accumulators as packed bf16 straight away (they are only ever MFMA operands): the two extra 16-register
        // fp32 copies this loop used to hold pushed the kernel 26 registers over its 256 (two waves per SIMD), and the spill traffic sat in
        // the loop (rocprof: 62 % of the wave cycles waiting)
        bf16x8 pbs[2], dsbs[2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 l4 = *(const f32x4*)(lse_s + 8 * g4 + 4 * half);
            const f32x4 d4 = *(const f32x4*)(dlt_s + 8 * g4 + 4 * half);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * g4 + j;
                const int qi = qb + 8 * g4 + 4 * half + j;
                const bool ok = qi < T && kpos < T && (!CAUSAL || kpos <= qi);
                const float pv = ok ? __builtin_amdgcn_exp2f(s[i] * sl - l4[j]) : 0.f;
                pbs[g4 >> 1][4 * (g4 & 1) + j] = (bf16)pv;
                dsbs[g4 >> 1][4 * (g4 & 1) + j] = (bf16)(pv * (dp[i] - d4[j]) * scale);        // dS[query][key]
            }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8 pb = pbs[st], dsb = dsbs[st];
#pragma unroll
            for (int d = 0; d < HD / 32; ++d) {
                const bf16x8 dot = tr_frag(do_tr, TS, 16 * st, 32 * d, lane);
                dva[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pb, dva[d], 0, 0, 0);      // dV^T += dO^T . P
                const bf16x8 qt = tr_frag(q_tr, TS, 16 * st, 32 * d, lane);
                dka[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, dsb, dka[d], 0, 0, 0);      // dK^T += Q^T . dS
            }
        }
    }
    if (!wave_active || kpos >= T) return;
    if (rope_tab) unrope<HD>(dka, rope_tab, kpos, half);
    bf16* okp = dk + ((long)b * T + kpos) * lddk + (long)hh * HD;
    bf16* ovp = dv + ((long)b * T + kpos) * lddv + (long)hh * HD;
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float a[4] = {dka[d][4 * g4], dka[d][4 * g4 + 1], dka[d][4 * g4 + 2], dka[d][4 * g4 + 3]};
            float c[4] = {dva[d][4 * g4], dva[d][4 * g4 + 1], dva[d][4 * g4 + 2], dva[d][4 * g4 + 3]};
            store_f<4>(okp + 32 * d + 8 * g4 + 4 * half, a);
            store_f<4>(ovp + 32 * d + 8 * g4 + 4 * half, c);
        }
}

template <int HD>
int launch_bwd(const void* q, const void* k, const void* v, const void* dout, const float* lse, float* delta, void* dq,
               void* dk, void* dv, int B, int T, int H, long ldq, long ldk, long ldv, long lddo, long lddq, long lddk, long lddv,
               float scale, int causal, hipStream_t st, int G, const void* o, long ldo, const float* rope_tab) {
    constexpr int NW = 4;
    const dim3 grid(av_cdiv(T, 32 * NW), H, B), gridkv(av_cdiv(T, 32 * NW), H / G, B), block(NW * 64);
    if (causal) {
        hipLaunchKernelGGL((attn_bwd_dq_mfma<HD, NW, true>), grid, block, 0, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)dout, lse, (float*)delta, (bf16*)dq, T, H, ldq, ldk, ldv, lddo, lddq, scale, G, (const bf16*)o, ldo, rope_tab);
        hipLaunchKernelGGL((attn_bwd_dkv_mfma<HD, NW, true>), gridkv, block, 0, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)dout, lse, delta, (bf16*)dk, (bf16*)dv, T, H, ldq, ldk, ldv, lddo, lddk, lddv, scale, G, rope_tab);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_mfma<HD, NW, false>), grid, block, 0, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)dout, lse, (float*)delta, (bf16*)dq, T, H, ldq, ldk, ldv, lddo, lddq, scale, G, (const bf16*)o, ldo, rope_tab);
        hipLaunchKernelGGL((attn_bwd_dkv_mfma<HD, NW, false>), gridkv, block, 0, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)dout, lse, delta, (bf16*)dk, (bf16*)dv, T, H, ldq, ldk, ldv, lddo, lddk, lddv, scale, G, rope_tab);
    }
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // namespace

int av_attention_bwd_mfma(const void* q, const void* k, const void* v, const void* dout, const float* lse, float* delta,
                          void* dq, void* dk, void* dv, int B, int T, int H, int hd, long ldq, long ldk, long ldv, long lddo,
                          long lddq, long lddk, long lddv, float scale, int causal, hipStream_t st, int G, const void* o, long ldo,
                          const float* rope_tab) {
    AV_CHECK_ARG(hd == 128 || hd == 64, "attention_bwd(mfma): head_dim %d unsupported", hd);
    if (hd == 128) return launch_bwd<128>(q, k, v, dout, lse, delta, dq, dk, dv, B, T, H, ldq, ldk, ldv, lddo, lddq, lddk, lddv, scale, causal, st, G, o, ldo, rope_tab);
    return launch_bwd<64>(q, k, v, dout, lse, delta, dq, dk, dv, B, T, H, ldq, ldk, ldv, lddo, lddq, lddk, lddv, scale, causal, st, G, o, ldo, rope_tab);
}
