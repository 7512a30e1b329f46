// Block-scaled fp8 (OCP MX: e4m3 elements, one E8M0 scale per 32 consecutive K elements) for BASELINE config 5
// ("Whisper-large-v3 + CLIP-ViT-L/14 -> Mistral-7B, fp8 MFMA").  The reference has no fp8 mode
// (src/clip_whisper/models/clip_whisper_model.py:164 only `use_fp16`): this is the MI355X rendering of that config's arithmetic --
// every frozen-weight projection of the forward pass runs on v_mfma_scale_f32_16x16x128_f8f6f4 (the only fp8 form that doubles the
// bf16 matrix rate: MI355X_MICROARCH.md "Matrix cores"), weights quantised once at load, activations quantised by a streaming kernel;
// accumulation, bias, activation, residual and the output stay fp32 -> bf16 exactly as in the bf16 path.
//
// Formats
//   q      uint8 [R, K] e4m3fn, row-major (K % 128 == 0)
//   scales "image" of uint32: for K-step t (128 elements), 64-row group rb, K-block fq (32 elements), lane row fr:
//              word(((t * RB + rb) * 4 + fq) * 16 + fr)  byte i = E8M0 exponent of row  row_of(layout, rb, i, fr),  K-block 4t + fq
//          so that lane (fr, fq) of a wave fetches, with ONE coalesced dword load, the four scale bytes of the four 16-row MFMA tiles it
//          feeds (the MFMA's OPSEL immediate then picks the byte: it is the same for all lanes, hence this byte-planar layout).
//          layout 0 (activations, MFMA "B" side):  row = 64 rb + 16 i + fr
//          layout 1 (weights, MFMA "A" side):      row = 128 (rb >> 1) + 32 (j >> 1) + 4 (j & 1) + 8 (fr >> 2) + (fr & 3),  j = 4 (rb & 1) + i
//              -- the column-interleaved fragment order of the 256x256 kernels (gemm.hip WP_BOFF): tiles 2p / 2p+1 give a lane 8
//              consecutive output columns, stored straight from the accumulators.
// Operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (measured: tools/ubench/mfma_scale_probe.hip): lane (r, G) (r = lane & 15 the
// row / column, G = lane >> 4) holds in VGPRs 0-3 the K elements [16 G, 16 G + 16) and in VGPRs 4-7 the elements [64 + 16 G, 64 + 16 G + 16)
// of the 128-element K-step -- the 16-byte chunks G and 4 + G of a 128-byte row -- and its scale register carries the E8M0 byte of MX
// block G = elements [32 G, 32 G + 32) of that row.  (A lane's scale therefore does NOT cover the lane's own 32 bytes: block 0 is the
// low halves of lane groups 0 and 1, block 2 their high halves.)
// E8M0 exponent rule (OCP MX v1.0 §6.3): e = floor(log2(amax)) - 8 (emax of e4m3), elements = RNE(x * 2^-e) saturated to +-448.
#include "common.h"
#include "avllm_internal.h"
#include "gemm_shared.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) int v4i;

__host__ __device__ __forceinline__ int mx_row(int layout, int rb, int i, int fr) {
    if (layout == 0) return 64 * rb + 16 * i + fr;
    const int j = 4 * (rb & 1) + i;
    return 128 * (rb >> 1) + 32 * (j >> 1) + 4 * (j & 1) + 8 * (fr >> 2) + (fr & 3);
}

// one thread = the four 32-element blocks (rows row_of(rb, 0..3, fr), K-block kb) that share one scale word
template <typename T>
__global__ __launch_bounds__(256) void mx_quant_kernel(const T* __restrict__ x, long ldx, int R, int K, uint8_t* __restrict__ q, long ldq,
                                                       uint32_t* __restrict__ simg, int RB, int layout) {
    const int nkb = K >> 5;
    const long total = (long)RB * 16 * nkb;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int kb = (int)(idx % nkb);
        const long rest = idx / nkb;
        const int fr = (int)(rest & 15), rb = (int)(rest >> 4);
        uint32_t word = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = mx_row(layout, rb, i, fr);
            if (row >= R) continue;
            float v[32];
            const T* xp = x + (long)row * ldx + kb * 32;
#pragma unroll
            for (int c = 0; c < 32; c += 8) {
                float t8[8];
                load_f<8>(xp + c, t8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c + j] = t8[j];
            }
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(v[j]));
            // floor(log2(amax)) from the exponent field (amax == 0 or subnormal -> smallest scale)
            int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
            e = e < -127 ? -127 : (e > 127 ? 127 : e);
            const float invs = __uint_as_float((uint32_t)(127 - e) << 23);         // 2^-e: e <= 120 for any finite amax, so the exponent field stays >= 7
            uint32_t pk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = fminf(fmaxf(v[4 * j + c] * invs, -448.f), 448.f);
                int r = 0;
                r = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], r, false);
                r = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], r, true);
                pk[j] = (uint32_t)r;
            }
            uint8_t* qp = q + (long)row * ldq + kb * 32;
            *(u32x4*)qp = (u32x4){pk[0], pk[1], pk[2], pk[3]};
            *(u32x4*)(qp + 16) = (u32x4){pk[4], pk[5], pk[6], pk[7]};
            word |= (uint32_t)(e + 127) << (8 * i);
        }
        simg[(((long)(kb >> 2) * RB + rb) * 4 + (kb & 3)) * 16 + fr] = word;
    }
}

struct F8Epi {
    void* C; long ldc; const void* bias; const void* R; long ldr; int act; int M, N;
};
struct GemmF8Args {
    const uint8_t* A; const uint8_t* B; const uint32_t* SA; const uint32_t* SB;
    long lda, ldb;
    int K, RBA, RBB, dbg;
    F8Epi e;
};

// ------------------------------------------------------------------------------------------ reference-grade kernel
// Any shape; fragments straight from global memory (no LDS reuse).  Defines the operand / scale / output mapping the fast kernel must
// reproduce, and serves the calls the fast kernel does not take (few tiles, ragged M handled by row clamps).
// Workgroup = 4 waves stacked in M: 256 rows x 128 columns; wave = 64 rows (4 row tiles) x 128 columns (8 column tiles).
__global__ __launch_bounds__(256) void gemm_f8_ref_kernel(GemmF8Args g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.y * 256 + wave * 64, cb = blockIdx.x, n0 = cb * 128;
    if (m0 >= g.e.M) return;
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int rbA = m0 >> 6;                                   // the wave's 64 rows are one natural scale group
    const uint8_t* ap[4]; const uint8_t* bp[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + 16 * i + fr;
        m = m < g.e.M ? m : g.e.M - 1;
        ap[i] = g.A + (long)m * g.lda + fq * 16;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int n = n0 + 32 * (j >> 1) + 4 * (j & 1) + 8 * (fr >> 2) + (fr & 3);
        n = n < g.e.N ? n : g.e.N - 1;
        bp[j] = g.B + (long)n * g.ldb + fq * 16;
    }
    for (int t = 0; t < g.K / 128; ++t) {
        const int sa = (int)g.SA[(((long)t * g.RBA + rbA) * 4 + fq) * 16 + fr];
        const int sb0 = (int)g.SB[(((long)t * g.RBB + 2 * cb) * 4 + fq) * 16 + fr], sb1 = (int)g.SB[(((long)t * g.RBB + 2 * cb + 1) * 4 + fq) * 16 + fr];
        v8i fa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v4i lo = *(const v4i*)(ap[i] + t * 128), hi = *(const v4i*)(ap[i] + t * 128 + 64);
            fa[i] = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const v4i lo = *(const v4i*)(bp[j] + t * 128), hi = *(const v4i*)(bp[j] + t * 128 + 64);
            const v8i fb = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const int sb = j < 4 ? sb0 : sb1;
#define AV_F8_MFMA(I, OB)                                                                                                               \
            acc[I][j] = (j & 3) == 0 ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa[I], acc[I][j], 0, 0, 0, sb, OB, sa)      \
                      : (j & 3) == 1 ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa[I], acc[I][j], 0, 0, 1, sb, OB, sa)      \
                      : (j & 3) == 2 ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa[I], acc[I][j], 0, 0, 2, sb, OB, sa)      \
                                     : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa[I], acc[I][j], 0, 0, 3, sb, OB, sa)
            AV_F8_MFMA(0, 0); AV_F8_MFMA(1, 1); AV_F8_MFMA(2, 2); AV_F8_MFMA(3, 3);
#undef AV_F8_MFMA
        }
    }
    // D[n][m]: lane (fr, fq) of tile (i, j): row m0 + 16 i + fr, columns n0 + 32 (j >> 1) + 4 (j & 1) + 8 fq + {0..3}
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 16 * i + fr;
        if (m >= g.e.M) continue;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int n = n0 + 32 * p + 8 * fq;
            float v[8] = {acc[i][2 * p][0], acc[i][2 * p][1], acc[i][2 * p][2], acc[i][2 * p][3],
                          acc[i][2 * p + 1][0], acc[i][2 * p + 1][1], acc[i][2 * p + 1][2], acc[i][2 * p + 1][3]};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (n + c >= g.e.N) break;
                float y = v[c] + (g.e.bias ? (float)((const bf16*)g.e.bias)[n + c] : 0.f);
                y = act_apply_fast(y, g.e.act);
                if (g.e.R) y += (float)((const bf16*)g.e.R)[(long)m * g.e.ldr + n + c];
                ((bf16*)g.e.C)[(long)m * g.e.ldc + n + c] = (bf16)y;
            }
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------ host API
// 64-row groups of a scale image: rows padded to a whole 256-row tile (the fast kernel fetches the words of all four groups of its tile)
static inline int mx_groups(int R) { return (R + 255) / 256 * 4; }
extern "C" size_t avllm_mx_scale_bytes(int32_t R, int32_t K) { return (size_t)(K / 128) * (size_t)mx_groups(R) * 4 * 16 * 4; }

int av_mx_quantize(const void* x, long ldx, int R, int K, void* q, long ldq, void* scales, int layout, int dtype, hipStream_t st) {
    AV_CHECK_ARG(x && q && scales && R > 0 && K > 0, "mx_quantize: null/empty");
    AV_CHECK_ARG(K % 128 == 0 && ldx % 8 == 0 && ldq % 16 == 0, "mx_quantize: K=%d must be a multiple of 128 (16-byte aligned rows)", K);
    AV_CHECK_ARG(layout == 0 || layout == 1, "mx_quantize: layout %d", layout);
    const int RB = mx_groups(R);
    const long total = (long)RB * 16 * (K / 32);
    long blocks = (total + 255) / 256;
    blocks = blocks > 65536 ? 65536 : blocks;
    if (dtype == AV_BF16) hipLaunchKernelGGL((mx_quant_kernel<bf16>), dim3(blocks), dim3(256), 0, st, (const bf16*)x, ldx, R, K, (uint8_t*)q, ldq, (uint32_t*)scales, RB, layout);
    else hipLaunchKernelGGL((mx_quant_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)x, ldx, R, K, (uint8_t*)q, ldq, (uint32_t*)scales, RB, layout);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

bool av_prof_enabled();
void av_prof_before(hipStream_t st);
void av_prof_after(hipStream_t st, double flops);
int av_gemm_f8_fast(const avllm_gemm_f8_desc* d, hipStream_t st, bool* taken);

int av_gemm_f8(const avllm_gemm_f8_desc* d, hipStream_t st) {
    AV_CHECK_ARG(d && d->A && d->B && d->SA && d->SB && (d->C || d->Cq), "gemm_f8: null operand");
    AV_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0 && d->K % 128 == 0, "gemm_f8: bad shape M=%d N=%d K=%d (K %% 128)", d->M, d->N, d->K);
    AV_CHECK_ARG(d->lda % 16 == 0 && d->ldb % 16 == 0 && d->ldc % 8 == 0 && (!d->R || d->ldr % 8 == 0), "gemm_f8: leading dims must keep 16-byte rows");
    const bool prof = av_prof_enabled();
    if (prof) av_prof_before(st);
    bool taken = false;
    AV_TRY(av_gemm_f8_fast(d, st, &taken));
    if (!taken && d->Cq)
        return av_set_error(AV_ERR_UNSUPPORTED, "gemm_f8: quantised output needs the persistent kernel (M > 128, >= 64 tiles, K >= 256, N %% 32 == 0, no residual)");
    if (!taken) {
        GemmF8Args g;
        g.A = (const uint8_t*)d->A; g.B = (const uint8_t*)d->B; g.SA = (const uint32_t*)d->SA; g.SB = (const uint32_t*)d->SB;
        g.lda = d->lda; g.ldb = d->ldb; g.K = d->K; g.RBA = mx_groups(d->M); g.RBB = mx_groups(d->N); g.dbg = 0;
        g.e.C = d->C; g.e.ldc = d->ldc; g.e.bias = d->bias; g.e.R = d->R; g.e.ldr = d->ldr; g.e.act = d->act; g.e.M = d->M; g.e.N = d->N;
        hipLaunchKernelGGL(gemm_f8_ref_kernel, dim3(av_cdiv(d->N, 128), av_cdiv(d->M, 256)), dim3(256), 0, st, g);
        AV_LAUNCH_CHECK();
    }
    if (prof) av_prof_after(st, 2.0 * d->M * (double)d->N * (double)d->K);
    return AV_OK;
}

extern "C" int avllm_mx_quantize(const void* x, int64_t ldx, int32_t R, int32_t K, void* q, int64_t ldq, void* scales, int32_t layout,
                                 int32_t dtype, void* stream) {
    return av_mx_quantize(x, ldx, R, K, q, ldq, scales, layout, dtype, (hipStream_t)stream);
}
extern "C" int avllm_gemm_f8(const avllm_gemm_f8_desc* d, void* stream) { return av_gemm_f8(d, (hipStream_t)stream); }

// ------------------------------------------------------------------------------------------ LayerNorm / RMSNorm -> MX e4m3 in one pass
// The producer of most fp8 projection inputs is a normalisation; writing it as bf16 and re-reading it in mx_quant_kernel is 4 bytes of HBM
// traffic per element for a 1-byte result.  One wave normalises a row from registers (bf16 in, fp32 statistics as norm.hip) and block-scales
// the fp32 result directly; a workgroup owns the 64 rows of one scale-image row group so that it can emit whole scale words.
namespace {
constexpr int NQ_MAXJ = 16;      // 16-byte chunks per lane: rows up to 64 * 8 * 16 = 8192 elements

template <bool RMS, int NQ_J>
__global__ __launch_bounds__(256) void norm_mxq_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, const bf16* __restrict__ b,
                                                       bf16* __restrict__ y, float* __restrict__ rstd_out, uint8_t* __restrict__ q, long ldq,
                                                       uint32_t* __restrict__ simg, int RB, long rows, int d, float eps) {
    __shared__ uint8_t sb[256][64];                       // [K block][row of the group]: E8M0 bytes (d <= 8192)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rb = blockIdx.x, nkb = d >> 5, nch = d >> 3;
    for (int rr = 0; rr < 16; ++rr) {
        const int rloc = 16 * wv + rr;                    // layout 0: byte i' = wave, fr' = rr
        const long row = (long)rb * 64 + rloc;
        if (row >= rows) {                                // wave-uniform
            for (int kb = lane; kb < nkb; kb += 64) sb[kb][rloc] = 0;
            continue;
        }
        const bf16* xr = x + row * d;
        float v[NQ_J][8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NQ_J; ++j) {
            const int c = lane + 64 * j;
            if (c < nch) {
                load_f<8>(xr + c * 8, v[j]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[j][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
            }
        }
        const float mean = RMS ? 0.f : wave_sum(s) / d;
        float qs = 0.f;
#pragma unroll
        for (int j = 0; j < NQ_J; ++j)
            if (lane + 64 * j < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float u = v[j][e] - mean; qs += u * u; }
            }
        const float rstd = rsqrtf(wave_sum(qs) / d + eps);
        if (rstd_out && lane == 0) rstd_out[row] = rstd;
#pragma unroll
        for (int j = 0; j < NQ_J; ++j) {
            const int c = lane + 64 * j;
            if (c < nch) {                                // d % 32 == 0: the 4 lanes of a 32-element block are in or out together
                float wv8[8], o[8];
                load_f<8>(w + c * 8, wv8);
                if (RMS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = wv8[e] * (v[j][e] * rstd);
                } else {
                    float bv8[8];
                    load_f<8>(b + c * 8, bv8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (v[j][e] - mean) * rstd * wv8[e] + bv8[e];
                }
                if (y) store_f<8>(y + row * d + c * 8, o);
                float amax = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(o[e]));
                amax = fmaxf(amax, __shfl_xor(amax, 1));
                amax = fmaxf(amax, __shfl_xor(amax, 2));
                int ex = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
                ex = ex < -127 ? -127 : (ex > 127 ? 127 : ex);
                const float invs = __uint_as_float((uint32_t)(127 - ex) << 23);
                float a8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] = fminf(fmaxf(o[e] * invs, -448.f), 448.f);
                int r0 = 0, r1 = 0;
                r0 = __builtin_amdgcn_cvt_pk_fp8_f32(a8[0], a8[1], r0, false);
                r0 = __builtin_amdgcn_cvt_pk_fp8_f32(a8[2], a8[3], r0, true);
                r1 = __builtin_amdgcn_cvt_pk_fp8_f32(a8[4], a8[5], r1, false);
                r1 = __builtin_amdgcn_cvt_pk_fp8_f32(a8[6], a8[7], r1, true);
                typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                *(u32x2*)(q + row * ldq + c * 8) = (u32x2){(uint32_t)r0, (uint32_t)r1};
                if ((lane & 3) == 0) sb[c >> 2][rloc] = (uint8_t)(ex + 127);
            }
        }
    }
    __syncthreads();
    // word ((t * RB + rb) * 4 + kb % 4) * 16 + fr' = bytes of rows 16 i' + fr', i' = 0..3, of K block kb = 4 t + kb % 4
    for (int idx = threadIdx.x; idx < nkb * 16; idx += 256) {
        const int kb = idx >> 4, fr = idx & 15;
        const uint32_t word = (uint32_t)sb[kb][fr] | ((uint32_t)sb[kb][16 + fr] << 8) | ((uint32_t)sb[kb][32 + fr] << 16) | ((uint32_t)sb[kb][48 + fr] << 24);
        simg[(((long)(kb >> 2) * RB + rb) * 4 + (kb & 3)) * 16 + fr] = word;
    }
}
}  // namespace

// y (optional bf16 copy), q / scales (layout 0) = block-scaled e4m3 of LayerNorm(x; w, b) (b != NULL) or RMSNorm(x; w) (b == NULL, rstd_out optional)
int av_norm_mxq(const void* x, const void* w, const void* b, void* y, float* rstd_out, void* q, long ldq, void* scales, long rows, int d, float eps,
                hipStream_t st) {
    AV_CHECK_ARG(x && w && q && scales && rows > 0, "norm_mxq: null/empty");
    AV_CHECK_ARG(d % 128 == 0 && d <= 64 * 8 * NQ_MAXJ && ldq >= d && ldq % 8 == 0, "norm_mxq: d=%d must be a multiple of 128 up to 8192", d);
    const int RB = mx_groups((int)rows);
#define NQ_LAUNCH(RMSV, J) hipLaunchKernelGGL((norm_mxq_kernel<RMSV, J>), dim3(RB), dim3(256), 0, st, (const bf16*)x, (const bf16*)w, (const bf16*)b, (bf16*)y, \
                                             rstd_out, (uint8_t*)q, ldq, (uint32_t*)scales, RB, rows, d, eps)
    const int nj = (d / 8 + 63) / 64;               // 16-byte chunks per lane
    if (b) { if (nj <= 2) NQ_LAUNCH(false, 2); else if (nj <= 4) NQ_LAUNCH(false, 4); else if (nj <= 8) NQ_LAUNCH(false, 8); else NQ_LAUNCH(false, 16); }
    else { if (nj <= 2) NQ_LAUNCH(true, 2); else if (nj <= 4) NQ_LAUNCH(true, 4); else if (nj <= 8) NQ_LAUNCH(true, 8); else NQ_LAUNCH(true, 16); }
#undef NQ_LAUNCH
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int avllm_norm_mxq(const void* x, const void* w, const void* b, void* y, float* rstd_out, void* q, int64_t ldq, void* scales, int64_t rows,
                              int32_t d, float eps, void* stream) {
    return av_norm_mxq(x, w, b, y, rstd_out, q, ldq, scales, rows, d, eps, (hipStream_t)stream);
}
