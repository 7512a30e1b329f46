// Flash attention forward for gfx950 (bf16 in, fp32 softmax/accumulate), head_dim 64 (Whisper / CLIP
// encoders, non-causal) and 128 (Llama, causal).  HF eager_attention_forward restated as a single pass:
// whisper :215-238, clip :259-277 (fp32 softmax), llama sdpa path.
//
// Layout choices (CDNA4): one wave owns 32 query rows; the workgroup (NW waves) shares K/V tiles of 64 keys
// staged in LDS.  Scores are computed TRANSPOSED, S^T = K.Q^T, with v_mfma_f32_32x32x16_bf16: the query
// index then lives on the lane, so the online-softmax running max / sum / rescale are lane-local scalars and
// the row reduction is 15 in-register max/adds + one cross-half shuffle.  The S^T accumulator registers are,
// after a pairwise bf16 convert, directly the B operand of O^T += V^T.P^T (cdna guide §3 "An accumulator tile
// as the next MFMA's operand"), so P never touches LDS.  V^T fragments come from the row-major V tile through
// ds_read_b64_tr_b16 (hardware transpose).  K rows are padded by 16 B and V rows by 64 B, which makes the
// ds_read_b128 row reads and the 4-row transposed reads bank-conflict free (MI355X_MICROARCH §LDS).
#include "common.h"
#include "avllm_internal.h"
#include <type_traits>
#include <cstdlib>

int av_attention_fwd_ref(const void* q, const void* k, const void* v, void* o, float* lse, int B, int Tq, int Tk, int H,
                         int hd, long ldq, long ldk, long ldv, long ldo, float scale, int causal, int dtype, hipStream_t st, int G);
int av_attention_delta(const void* o, const void* dout, float* delta, int B, int T, int H, int hd, long ldo, long lddo, int dtype, hipStream_t st);
int av_attention_bwd_ref(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                         void* dq, void* dk, void* dv, int B, int T, int H, int hd, long ldq, long ldk, long ldv, long lddo,
                         long lddq, long lddk, long lddv, float scale, int causal, int dtype, hipStream_t st, int G);
int av_attention_bwd_mfma(const void* q, const void* k, const void* v, const void* dout, const float* lse, float* delta,
                          void* dq, void* dk, void* dv, int B, int T, int H, int hd, long ldq, long ldk, long ldv, long lddo,
                          long lddq, long lddk, long lddv, float scale, int causal, hipStream_t st, int G, const void* o, long ldo, const float* rope_tab);

namespace {

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
__device__ __forceinline__ float max3f(float a, float x, float y) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, x), y); }

template <int HD, int NW, bool CAUSAL>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_mfma(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                         const bf16* __restrict__ v, bf16* __restrict__ o, float* __restrict__ lse,
                                                         int Tq, int Tk, int H, long ldq, long ldk, long ldv, long ldo,
                                                         float scale_log2e, int G) {
    constexpr int KS = HD * 2 + 16;      // K tile row stride (bytes)
    constexpr int VS = HD * 2 + 64;      // V tile row stride (bytes)
    constexpr int CPR = HD / 8;          // 16-byte chunks per row
    constexpr int NT = NW * 64;
    // short non-causal sequences (CLIP: 197 tokens, 7 waves): the WHOLE K/V of the (item, head) is staged once (224 rows,
    // 75 KiB -> two workgroups per CU) and the key loop runs without any further barrier; otherwise 64-key tiles.
    constexpr bool ONESHOT = (HD == 64 && NW == 7 && !CAUSAL);
    constexpr int LROWS = ONESHOT ? 224 : 64;
    // DB (long sequences at head_dim 64: Whisper, T = 1500, 24 tiles): the K/V tile buffers are doubled, tile t+1 is parked in the other buffer
    // right after the barrier that opens iteration t and ONE barrier per tile remains (two with a single buffer: "previous tile consumed" and
    // "this tile stored").  Two buffers at head_dim 128 would pass the 64 KiB of static LDS.
    constexpr bool DB = (HD == 64 && NW == 4);
    constexpr int NBUF = DB ? 2 : 1;
    __shared__ __attribute__((aligned(16))) char k_lds_[NBUF * LROWS * KS];
    __shared__ __attribute__((aligned(16))) char v_lds_[NBUF * LROWS * VS];
    char* k_lds = k_lds_;
    char* v_lds = v_lds_;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int hh = blockIdx.y, b = blockIdx.z, hk = hh / G;     // hk: the key/value head this query head reads (grouped-query attention)
    const int q0 = (blockIdx.x * NW + w) * 32;
    const int off = Tk - Tq;                                   // causal: query t sees keys <= t + off
    const int qpos = q0 + r;
    const int qrow = qpos < Tq ? qpos : Tq - 1;

    // Q fragments: B operand of S^T = K.Q^T  -> lane holds Q[query r][16ks + 8half .. +8]
    bf16x8 qf[HD / 16];
    {
        const bf16* qp = q + ((long)b * Tq + qrow) * ldq + (long)hh * HD + 8 * half;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) qf[ks] = *(const bf16x8*)(qp + 16 * ks);
    }
    f32x16 oacc[HD / 32];
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int blk_qmax = min(Tq - 1, (int)(blockIdx.x * NW + NW) * 32 - 1);
    const int k_end = CAUSAL ? min(Tk, blk_qmax + off + 1) : Tk;      // keys [0,k_end) are needed by this block
    const bool wave_active = q0 < Tq;
    const int wave_kmax = CAUSAL ? min(Tk - 1, min(Tq - 1, q0 + 31) + off) : Tk - 1;

    // K/V tiles are staged global -> registers -> LDS; the loads of tile t+1 are issued right after tile t has been
    // written to LDS, so their latency runs under tile t's MFMA/softmax work (async-STAGE split, cdna guide T14).
    constexpr int NCH = (64 * CPR + NT - 1) / NT;
    // only where the extra 8*NCH VGPRs do not cost a resident workgroup: long non-causal sequences (Whisper, T=1500)
    constexpr bool PREFETCH = (HD == 64 && NW == 4);
    u32x4 kreg[NCH], vreg[NCH];
    auto prefetch = [&](int kb0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NT;
            const int row = c / CPR, ch = c % CPR;
            const int key = kb0 + row;
            kreg[i] = (u32x4){0u, 0u, 0u, 0u}; vreg[i] = (u32x4){0u, 0u, 0u, 0u};
            if (c < 64 * CPR && key < Tk) {
                kreg[i] = *(const u32x4*)(k + ((long)b * Tk + key) * ldk + (long)hk * HD + ch * 8);
                vreg[i] = *(const u32x4*)(v + ((long)b * Tk + key) * ldv + (long)hk * HD + ch * 8);
            }
        }
    };
    if (ONESHOT) {
        // every global load of the K/V image is issued before the first LDS store waits on one: a single exposed latency
        constexpr int NONE = (LROWS * CPR + NT - 1) / NT;
        u32x4 kall[NONE], vall[NONE];
#pragma unroll
        for (int i = 0; i < NONE; ++i) {
            const int c = tid + i * NT, row = c / CPR, ch = c % CPR;
            kall[i] = (u32x4){0u, 0u, 0u, 0u}; vall[i] = (u32x4){0u, 0u, 0u, 0u};
            if (c < LROWS * CPR && row < Tk) {
                kall[i] = *(const u32x4*)(k + ((long)b * Tk + row) * ldk + (long)hk * HD + ch * 8);
                vall[i] = *(const u32x4*)(v + ((long)b * Tk + row) * ldv + (long)hk * HD + ch * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < NONE; ++i) {
            const int c = tid + i * NT, row = c / CPR, ch = c % CPR;
            if (c < LROWS * CPR) {
                *(u32x4*)(k_lds + row * KS + ch * 16) = kall[i];
                *(u32x4*)(v_lds + row * VS + ch * 16) = vall[i];
            }
        }
        __syncthreads();
    }
    auto park = [&](int buf) {                      // the staged registers -> tile buffer buf
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NT;
            if (c < 64 * CPR) {
                const int row = c / CPR, ch = c % CPR;
                *(u32x4*)(k_lds_ + buf * LROWS * KS + row * KS + ch * 16) = kreg[i];
                *(u32x4*)(v_lds_ + buf * LROWS * VS + row * VS + ch * 16) = vreg[i];
            }
        }
    };
    if (PREFETCH) prefetch(0);
    if (DB) { park(0); if (64 < k_end) prefetch(64); }
    for (int kb = 0; kb < k_end; kb += 64) {
        const int lrow0 = ONESHOT ? kb : 0;        // first LDS row of this 64-key step
        if (DB) {
            const int buf = (kb >> 6) & 1;
            __syncthreads();                       // tile kb/64 is stored (previous iteration) and the other buffer has been read by everybody
            if (kb + 64 < k_end) { park(buf ^ 1); if (kb + 128 < k_end) prefetch(kb + 128); }
            k_lds = k_lds_ + buf * LROWS * KS; v_lds = v_lds_ + buf * LROWS * VS;
        } else if (!ONESHOT) {
            if (!PREFETCH) prefetch(kb);
            __syncthreads();                       // previous tile fully consumed
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tid + i * NT;
                if (c < 64 * CPR) {
                    const int row = c / CPR, ch = c % CPR;
                    *(u32x4*)(k_lds + row * KS + ch * 16) = kreg[i];
                    *(u32x4*)(v_lds + row * VS + ch * 16) = vreg[i];
                }
            }
            __syncthreads();
            if (PREFETCH && kb + 64 < k_end) prefetch(kb + 64);
        }
        if (!wave_active || kb > wave_kmax) continue;      // wave-uniform

        const bool two = (kb + 32 <= wave_kmax) && (kb + 32 < Tk);   // second 32-key sub-block has visible keys
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            const bf16x8 kf0 = *(const bf16x8*)(k_lds + (lrow0 + r) * KS + (2 * ks + half) * 16);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[ks], s0, 0, 0, 0);
        }
        if (two) {
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) {
                const bf16x8 kf1 = *(const bf16x8*)(k_lds + (lrow0 + 32 + r) * KS + (2 * ks + half) * 16);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[ks], s1, 0, 0, 0);
            }
        }
        // mask (edge / diagonal blocks only) + block max on the RAW scores; the scale rides in the exp2's fma
        const int klim = CAUSAL ? min(Tk - 1, qpos + off) : Tk - 1;
        const bool need_mask = (kb + 64 > Tk) || !two || (CAUSAL && kb + 63 > q0 + off);
        float mloc = -INFINITY;
        if (need_mask) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key0 = kb + acc_row(i, half);
                s0[i] = key0 <= klim ? s0[i] : -INFINITY;
                s1[i] = (two && key0 + 32 <= klim) ? s1[i] : -INFINITY;
            }
        }
        // IEEE-2019 maximum (v_maximum3_f32 on gfx950): fmaxf canonicalises each operand with a v_max x, x first (three instructions per two
        // scores instead of one).  It has to stay a builtin: the scores are MFMA results, and the compiler pads the MFMA -> VALU read hazard
        // only for instructions it can see (an inline-asm v_max3 here read scores before the matrix pipe had written them).
#pragma unroll
        for (int i = 0; i < 16; ++i) mloc = max3f(mloc, s0[i], s1[i]);
        mloc = __builtin_elementwise_maximum(mloc, __shfl_xor(mloc, 32));
        mloc *= scale_log2e;
        const float m_new = __builtin_elementwise_maximum(m_run, mloc);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s0[i] = __builtin_amdgcn_exp2f(fmaf(s0[i], scale_log2e, -m_use));
            s1[i] = __builtin_amdgcn_exp2f(fmaf(s1[i], scale_log2e, -m_use));
            psum += s0[i] + s1[i];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < HD / 32; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;

        // O^T += V^T . P^T   (k = keys; 2 k-steps of 16 per 32-key sub-block)
        const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            if (sb == 1 && !two) break;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (bf16)(sb == 0 ? s0[8 * s + j] : s1[8 * s + j]);
                const int krow = lrow0 + 32 * sb + 16 * s + 4 * half + (i16 >> 2);
#pragma unroll
                for (int d = 0; d < HD / 32; ++d) {
                    const int col = 32 * d + 16 * (g & 1) + 4 * (i16 & 3);
                    const char* a0 = v_lds + krow * VS + col * 2;
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0));
                    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0 + 8 * VS));
                    typedef __attribute__((ext_vector_type(8))) short short8v;
                    const short8v both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x8 vt = __builtin_bit_cast(bf16x8, both);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt, pb, oacc[d], 0, 0, 0);
                }
            }
        }
    }
    if (ONESHOT) {
        // Output through LDS: the MFMA layout gives every lane 8-byte pieces of 32 different rows (64 scattered stores per
        // instruction); parked in the (now free) K image as bf16 rows, the wave's 32 x HD tile leaves as whole 16-byte row chunks.
        __syncthreads();                                       // every wave is done reading K/V
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
        char* ot = k_lds + w * 32 * KS;
#pragma unroll
        for (int d = 0; d < HD / 32; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float vals[4] = {oacc[d][4 * g4] * inv, oacc[d][4 * g4 + 1] * inv, oacc[d][4 * g4 + 2] * inv, oacc[d][4 * g4 + 3] * inv};
                store_f<4>((bf16*)(ot + r * KS) + 32 * d + 8 * g4 + 4 * half, vals);
            }
        if (lse && half == 0 && qpos < Tq) lse[((long)b * H + hh) * Tq + qpos] = (m_run + log2f(l_tot)) * 0.69314718055994531f;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 32 * CPR / 64; ++i) {
            const int c = lane + i * 64, row = c / CPR, ch = c % CPR;
            if (q0 + row < Tq)
                *(u32x4*)(o + ((long)b * Tq + q0 + row) * ldo + (long)hh * HD + ch * 8) = *(const u32x4*)(ot + row * KS + ch * 16);
        }
        return;
    }
    if (!wave_active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (qpos < Tq) {
        bf16* op = o + ((long)b * Tq + qpos) * ldo + (long)hh * HD;
#pragma unroll
        for (int d = 0; d < HD / 32; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float vals[4] = {oacc[d][4 * g4] * inv, oacc[d][4 * g4 + 1] * inv, oacc[d][4 * g4 + 2] * inv, oacc[d][4 * g4 + 3] * inv};
                store_f<4>(op + 32 * d + 8 * g4 + 4 * half, vals);
            }
        if (lse && half == 0) lse[((long)b * H + hh) * Tq + qpos] = (m_run + log2f(l_tot)) * 0.69314718055994531f;
    }
}

// ------------------------------------------------------------------------------------------ short non-causal sequences (CLIP)
// One workgroup = one (frame, head); the whole K and V of the head (<= 16 NKB keys) are staged once.  A wave takes 16 queries at a time
// and keeps ALL their scores in registers (4 NKB fp32 per lane), so the softmax is two exact passes over registers -- no running max,
// no rescale of the output accumulator, no alpha exponentials -- and every product is v_mfma_f32_16x16x32_bf16:
//   S^T[key][query] = K . Q^T      A = K rows from LDS (ds_read_b128, 160-byte rows: conflict free), B = the wave's Q rows (registers)
//   row max          over the lane's 4 NKB registers, then across the four lane groups that share a query (2 cross-lane steps)
//   P = exp2(S * scale*log2e - max)                                                        (the only transcendental per score)
//   l   += 1^T . P^T               the row SUM is one more MFMA with an all-ones A operand instead of 4 NKB VALU adds
//   O^T += V^T . P^T               P's registers, converted pairwise to bf16, ARE the B operand (key order permuted inside a 32-key
//                                  step: elements 0..3 = keys 4g..4g+3 of the first 16-key block, 4..7 = the same of the second);
//                                  V^T fragments come from the row-major V image through ds_read_b64_tr_b16 in that same order.
// 197 tokens (ViT-B/16) are 13 blocks of 16 in both directions (208 padded rows instead of the 224 of a 32-row tiling); 7 waves x 2
// query blocks.  Output leaves through a 16x32 bf16 scratch per wave as 16-byte row chunks.  HF:models/clip/modeling_clip.py:297-335.
// QOUT: the output leaves block-scaled to e4m3 (OCP MX: one E8M0 per 32 consecutive features, fp8.hip mx_quant_kernel's rule applied to the
// bf16-rounded values, i.e. bit-identical to quantising the bf16 output in a separate pass) as codes oq [rows, ldoq] + the layout-0 scale
// image osc of an avllm_gemm_f8 A operand: the fp8 out-projection reads it directly and the [M, d] bf16 attention output never exists
// (1.6 GB written + read back per ViT-L/14 layer at 750 frames x 4 clips).  A 32-feature block of a row = the 4 lanes of one row in the
// write-out below; the scale byte of global row m, column block cb sits at ((((cb >> 2) * RB + (m >> 6)) * 4 + (cb & 3)) * 16 + (m & 15)) * 4 + ((m >> 4) & 3).
// FULLK: the sequence uses every key block (nkb == NKB, e.g. CLIP's 197 tokens = 13 blocks), decided at launch: as a run-time branch the
// compiler hoisted the OTHER form's per-block "is this key inside T" masks above it (52 compares, 96 v_writelane of spilled scalar masks:
// as many instructions as a whole query block, paid by every wave).
template <int NKB, int NW, bool QOUT, bool FULLK>
__global__ __launch_bounds__(NW * 64) void attn_fwd_short(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                          bf16* __restrict__ o, float* __restrict__ lse, int T, int H, long ldq, long ldk,
                                                          long ldv, long ldo, float scale_log2e, int G, uint8_t* __restrict__ oq, long ldoq,
                                                          uint8_t* __restrict__ osc, int RBo) {
    constexpr int HD = 64, KS = 160, VS = 160, OS = 80;
    constexpr int NPV = (NKB + 1) / 2, KROWS = NKB * 16, VROWS = NPV * 32, NT = NW * 64;
    constexpr int NQ = (NKB + NW - 1) / NW;                       // query blocks per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* v_lds = smem + KROWS * KS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    char* o_lds = smem + KROWS * KS + VROWS * VS + w * 16 * OS;
    const int fr = lane & 15, fq = lane >> 4;
    const int hh = blockIdx.x, b = blockIdx.y, hk = hh / G;
    const int nkb = (T + 15) >> 4;

    // ---- stage K (rows < KROWS) and V (rows < VROWS), zero past T; every global load is issued before the first LDS store
    constexpr int NCH = (VROWS * 8 + NT - 1) / NT;
    u32x4 kreg[NCH], vreg[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NT, row = c >> 3, ch = c & 7;
        kreg[i] = (u32x4){0u, 0u, 0u, 0u}; vreg[i] = (u32x4){0u, 0u, 0u, 0u};
        if (row < T) {
            kreg[i] = *(const u32x4*)(k + ((long)b * T + row) * ldk + (long)hk * HD + ch * 8);
            vreg[i] = *(const u32x4*)(v + ((long)b * T + row) * ldv + (long)hk * HD + ch * 8);
        }
    }
    // Q fragments of this wave's query blocks: B operand, lane holds Q[query fr][32 ks + 8 fq .. +7]
    bf16x8 qf[NQ][2];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int qrow = (w + NW * i) * 16 + fr;
        qrow = qrow < T ? qrow : T - 1;
        const bf16* qp = q + ((long)b * T + qrow) * ldq + (long)hh * HD + 8 * fq;
        qf[i][0] = *(const bf16x8*)qp;
        qf[i][1] = *(const bf16x8*)(qp + 32);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NT, row = c >> 3, ch = c & 7;
        if (row < KROWS) *(u32x4*)(k_lds + row * KS + ch * 16) = kreg[i];
        if (row < VROWS) *(u32x4*)(v_lds + row * VS + ch * 16) = vreg[i];
    }
    __syncthreads();

    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
    asm volatile("" : "+v"(ones));           // opaque: left as a constant it was re-materialised with four v_mov before every row-sum MFMA (28 per query block)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, g = lane >> 4;
    // The kernel is bound by its instruction stream (profiles/r02_experiments.txt), so the per-score work is trimmed: v_maximum3_f32 through
    // max3f (fmaxf canonicalises every operand with a v_max x, x first: three instructions per two scores instead of one), and a FULL form for
    // sequences that use every key block (CLIP: 197 tokens = 13 blocks) without the per-block "is this block inside T" branches.
    auto max3 = [](float a, float x, float y) __attribute__((always_inline)) { return max3f(a, x, y); };
    auto qblock = [&](int qi, auto fullc) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(fullc)::value;             // nkb == NKB
        const int qb = w + NW * qi;
        const int q0 = qb * 16;
        // ---- scores: S^T block kb = keys 16kb .. +15, this lane: keys 16kb + 4fq + {0..3} of query q0 + fr
        f32x4 sc[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            sc[kb] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (FULL || kb < nkb) {
                const char* kr = k_lds + (kb * 16 + fr) * KS + fq * 16;
                const bf16x8 k0 = *(const bf16x8*)kr, k1 = *(const bf16x8*)(kr + 64);
                f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qi][0], zero4, 0, 0, 0);
                sc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qi][1], a, 0, 0, 0);
            }
        }
        if (T & 15) {                                             // keys past T in the last block
            if constexpr (FULL) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sc[NKB - 1][i] = ((NKB - 1) * 16 + 4 * fq + i < T) ? sc[NKB - 1][i] : -INFINITY;
            } else {
                const int kb = nkb - 1;
#pragma unroll
                for (int x = 0; x < NKB; ++x)
                    if (x == kb) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) sc[x][i] = (x * 16 + 4 * fq + i < T) ? sc[x][i] : -INFINITY;
                    }
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) { m = max3(m, sc[kb][0], sc[kb][1]); m = max3(m, sc[kb][2], sc[kb][3]); }
        m = max3(m, __shfl_xor(m, 16), m);
        m = max3(m, __shfl_xor(m, 32), m);
        const float msc = m * scale_log2e;
        // ---- P, row sum and O^T
        f32x4 lacc = zero4, oacc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
        for (int st = 0; st < NPV; ++st) {
            if (FULL || 2 * st < nkb) {
                bf16x8 pb;
                const bool second = 2 * st + 1 < NKB;            // compile-time after unrolling: the last step of an odd NKB has one block
                const f32x4 s1 = sc[second ? 2 * st + 1 : 2 * st];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pb[i] = (bf16)__builtin_amdgcn_exp2f(fmaf(sc[2 * st][i], scale_log2e, -msc));
                    pb[4 + i] = second ? (bf16)__builtin_amdgcn_exp2f(fmaf(s1[i], scale_log2e, -msc)) : (bf16)0.f;
                }
                lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb, lacc, 0, 0, 0);
                const char* vr = v_lds + (32 * st + 4 * g + (i16 >> 2)) * VS + (4 * (i16 & 3)) * 2;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(vr + t * 32));
                    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(vr + t * 32 + 16 * VS));
                    typedef __attribute__((ext_vector_type(8))) short short8v;
                    const short8v both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    oacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, both), pb, oacc[t], 0, 0, 0);
                }
            }
        }
        // every row of lacc is the row sum: lane (fr, *) holds l of query q0 + fr
        const float l = lacc[0], inv = 1.0f / l;
        if (lse && fq == 0 && q0 + fr < T) lse[((long)b * H + hh) * T + q0 + fr] = (msc + log2f(l)) * 0.69314718055994531f;
        // ---- output: O^T[16t + 4fq + i][query fr] -> 16 x 32 bf16 scratch (two halves of the head dim) -> 16-byte row chunks
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const f32x4 a = oacc[2 * hf + tt];
                float vals[4] = {a[0] * inv, a[1] * inv, a[2] * inv, a[3] * inv};
                store_f<4>((bf16*)(o_lds + fr * OS) + 16 * tt + 4 * fq, vals);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS traffic is ordered; pins the compiler's order
            const int row = lane >> 2, ch = lane & 3;
            const u32x4 val = *(const u32x4*)(o_lds + row * OS + ch * 16);
            if constexpr (QOUT) {
                const bf16x8 vb = __builtin_bit_cast(bf16x8, val);
                float f[8], amax = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) { f[c] = (float)vb[c]; amax = fmaxf(amax, fabsf(f[c])); }
                amax = fmaxf(amax, __shfl_xor(amax, 1));
                amax = fmaxf(amax, __shfl_xor(amax, 2));
                int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;              // floor(log2 amax) - 8 (fp8.hip mx_quant_kernel, oracle/mxfp8.py)
                e = e < -127 ? -127 : (e > 127 ? 127 : e);
                const float invs = __uint_as_float((uint32_t)(127 - e) << 23);
#pragma unroll
                for (int c = 0; c < 8; ++c) f[c] = fminf(fmaxf(f[c] * invs, -448.f), 448.f);
                int r0 = 0, r1 = 0;
                r0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], r0, false);
                r0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], r0, true);
                r1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], r1, false);
                r1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], r1, true);
                if (q0 + row < T) {
                    const long m = (long)b * T + q0 + row;
                    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                    *(u32x2*)(oq + m * ldoq + (long)hh * HD + hf * 32 + ch * 8) = (u32x2){(uint32_t)r0, (uint32_t)r1};
                    if (ch == 0) {
                        const int cb = hh * (HD / 32) + hf;
                        osc[((((long)(cb >> 2) * RBo + (m >> 6)) * 4 + (cb & 3)) * 16 + (m & 15)) * 4 + ((m >> 4) & 3)] = (uint8_t)(e + 127);
                    }
                }
            } else {
                if (q0 + row < T) *(u32x4*)(o + ((long)b * T + q0 + row) * ldo + (long)hh * HD + hf * 32 + ch * 8) = val;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    if constexpr (FULLK) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi)
            if (w + NW * qi < NKB) qblock(qi, std::true_type{});
    } else {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi)
            if (w + NW * qi < nkb) qblock(qi, std::false_type{});
    }
}

template <int NKB, int NW>
int launch_fwd_short(const void* q, const void* k, const void* v, void* o, float* lse, int B, int T, int H, long ldq, long ldk, long ldv,
                     long ldo, float scale, hipStream_t st, int G, void* oq = nullptr, long ldoq = 0, void* osc = nullptr) {
    constexpr int LDS = NKB * 16 * 160 + ((NKB + 1) / 2) * 32 * 160 + NW * 16 * 80;
    static bool attr[64] = {};
    int dev = 0;
    AV_HIP(hipGetDevice(&dev));
    if (!attr[dev & 63]) {
        AV_HIP(hipFuncSetAttribute((const void*)attn_fwd_short<NKB, NW, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        AV_HIP(hipFuncSetAttribute((const void*)attn_fwd_short<NKB, NW, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        AV_HIP(hipFuncSetAttribute((const void*)attn_fwd_short<NKB, NW, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        AV_HIP(hipFuncSetAttribute((const void*)attn_fwd_short<NKB, NW, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr[dev & 63] = true;
    }
    const int RBo = (int)(((long)B * T + 255) / 256 * 4);            // mx_groups(rows) of fp8.hip: 64-row groups, padded to whole 256-row tiles
    const bool fullk = ((T + 15) >> 4) == NKB;
#define AV_SHORT(QV, FV, OQ, LDOQ, OSC, RB) hipLaunchKernelGGL((attn_fwd_short<NKB, NW, QV, FV>), dim3(H, B), dim3(NW * 64), LDS, st, (const bf16*)q, (const bf16*)k, \
        (const bf16*)v, (bf16*)o, lse, T, H, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f, G, OQ, LDOQ, OSC, RB)
    if (oq) { if (fullk) AV_SHORT(true, true, (uint8_t*)oq, ldoq, (uint8_t*)osc, RBo); else AV_SHORT(true, false, (uint8_t*)oq, ldoq, (uint8_t*)osc, RBo); }
    else { if (fullk) AV_SHORT(false, true, nullptr, 0, nullptr, 0); else AV_SHORT(false, false, nullptr, 0, nullptr, 0); }
#undef AV_SHORT
    AV_LAUNCH_CHECK();
    return AV_OK;
}

template <int HD, int NW>
int launch_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int Tq, int Tk, int H, long ldq,
               long ldk, long ldv, long ldo, float scale, int causal, hipStream_t st, int G) {
    const dim3 grid(av_cdiv(Tq, 32 * NW), H, B), block(NW * 64);
    const float sl = scale * 1.4426950408889634f;
    if (causal) hipLaunchKernelGGL((attn_fwd_mfma<HD, NW, true>), grid, block, 0, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, lse, Tq, Tk, H, ldq, ldk, ldv, ldo, sl, G);
    else hipLaunchKernelGGL((attn_fwd_mfma<HD, NW, false>), grid, block, 0, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, lse, Tq, Tk, H, ldq, ldk, ldv, ldo, sl, G);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // namespace

int av_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int Tq, int Tk, int H,
                     int hd, long ldq, long ldk, long ldv, long ldo, float scale, int causal, int dtype, int impl,
                     hipStream_t st, int kv_heads) {
    AV_CHECK_ARG(q && k && v && o && B > 0 && Tq > 0 && Tk > 0 && H > 0, "attention_fwd: bad args");
    if (kv_heads <= 0) kv_heads = H;
    AV_CHECK_ARG(H % kv_heads == 0, "attention_fwd: %d query heads are not a multiple of %d key/value heads", H, kv_heads);
    const int G = H / kv_heads;
    AV_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "attention_fwd: row strides must be multiples of 8");
    AV_CHECK_ARG(!causal || Tk >= Tq, "attention_fwd: causal needs Tk >= Tq");
    if (impl == 1 || dtype == AV_F32 || (hd != 64 && hd != 128))
        return av_attention_fwd_ref(q, k, v, o, lse, B, Tq, Tk, H, hd, ldq, ldk, ldv, ldo, scale, causal, dtype, st, G);
    if (hd == 64) {
        // short non-causal self-attention (CLIP: 197 tokens, ViT-L/14: 257): whole-sequence scores in registers, exact two-pass softmax
        const bool short_off = av_knob(AV_KNOB_ATTN_SHORT) == 0;
        if (!causal && Tq == Tk && !short_off && B <= 65535) {
            if (Tq <= 208) return launch_fwd_short<13, 7>(q, k, v, o, lse, B, Tq, H, ldq, ldk, ldv, ldo, scale, st, G);
            if (Tq <= 272) return launch_fwd_short<17, 9>(q, k, v, o, lse, B, Tq, H, ldq, ldk, ldv, ldo, scale, st, G);
        }
        // short sequences (CLIP: 197 tokens): one workgroup covers the whole sequence with 7 waves
        if (Tq <= 224 && Tq > 128 && Tk <= 224 && !causal) return launch_fwd<64, 7>(q, k, v, o, lse, B, Tq, Tk, H, ldq, ldk, ldv, ldo, scale, causal, st, G);
        return launch_fwd<64, 4>(q, k, v, o, lse, B, Tq, Tk, H, ldq, ldk, ldv, ldo, scale, causal, st, G);
    }
    return launch_fwd<128, 4>(q, k, v, o, lse, B, Tq, Tk, H, ldq, ldk, ldv, ldo, scale, causal, st, G);
}

// Non-causal self-attention of short sequences with the output block-scaled to e4m3 in the epilogue (attn_fwd_short<.., QOUT>): codes
// oq [B*T, ldoq] + the layout-0 scale image an avllm_gemm_f8 A operand takes.  Only the shapes the one-pass kernel covers.
bool av_attention_fwd_mxq_ok(int B, int T, int H, int hd, int dtype, int kv_heads) {
    return dtype == AV_BF16 && hd == 64 && T <= 272 && B <= 65535 && (kv_heads <= 0 || kv_heads == H) && (H * hd) % 128 == 0 && av_knob(AV_KNOB_ATTN_SHORT) != 0 &&
           !av_knob(AV_KNOB_F8_UNFUSED_QUANT);
}
int av_attention_fwd_mxq(const void* q, const void* k, const void* v, void* oq, long ldoq, void* osc, int B, int T, int H, int hd, long ldq, long ldk,
                         long ldv, float scale, hipStream_t st) {
    AV_CHECK_ARG(q && k && v && oq && osc && av_attention_fwd_mxq_ok(B, T, H, hd, AV_BF16, H) && ldoq % 16 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0,
                 "attention_fwd_mxq: bad args (bf16, head_dim 64, T <= 272, 16-byte aligned code rows)");
    if (T <= 208) return launch_fwd_short<13, 7>(q, k, v, nullptr, nullptr, B, T, H, ldq, ldk, ldv, 0, scale, st, 1, oq, ldoq, osc);
    return launch_fwd_short<17, 9>(q, k, v, nullptr, nullptr, B, T, H, ldq, ldk, ldv, 0, scale, st, 1, oq, ldoq, osc);
}
extern "C" int avllm_attention_fwd_mxq(const void* q, const void* k, const void* v, void* oq, int64_t ldoq, void* scales, int32_t B, int32_t T, int32_t H,
                                       int32_t hd, int64_t ldq, int64_t ldk, int64_t ldv, float scale, void* stream) {
    return av_attention_fwd_mxq(q, k, v, oq, ldoq, scales, B, T, H, hd, ldq, ldk, ldv, scale, (hipStream_t)stream);
}

bool av_attention_bwd_fuses_rope(int dtype, int hd, int impl) { return impl == 0 && dtype == AV_BF16 && (hd == 128 || hd == 64); }

int av_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                     void* dq, void* dk, void* dv, float* delta_ws, int B, int T, int H, int hd, long ldq, long ldk,
                     long ldv, long ldo, long lddq, long lddk, long lddv, float scale, int causal, int dtype, int impl,
                     hipStream_t st, int kv_heads, const float* rope_tab) {
    AV_CHECK_ARG(q && k && v && o && dout && lse && dq && dk && dv && delta_ws, "attention_bwd: null");
    AV_CHECK_ARG(!rope_tab || av_attention_bwd_fuses_rope(dtype, hd, impl), "attention_bwd: the fused inverse RoPE exists in the bf16 MFMA kernels only");
    if (kv_heads <= 0) kv_heads = H;
    AV_CHECK_ARG(H % kv_heads == 0, "attention_bwd: %d query heads are not a multiple of %d key/value heads", H, kv_heads);
    const int G = H / kv_heads;
    // dO shares O's row stride.  The MFMA dQ kernel computes delta itself (and leaves it in delta_ws for the dK/dV kernel)
    if (impl == 0 && dtype == AV_BF16 && (hd == 128 || hd == 64))
        return av_attention_bwd_mfma(q, k, v, dout, lse, delta_ws, dq, dk, dv, B, T, H, hd, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, causal, st, G, o, ldo, rope_tab);
    AV_TRY(av_attention_delta(o, dout, delta_ws, B, T, H, hd, ldo, ldo, dtype, st));
    return av_attention_bwd_ref(q, k, v, dout, lse, delta_ws, dq, dk, dv, B, T, H, hd, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, causal, dtype, st, G);
}
