/*
 * avllm.h -- C ABI of libavllm.so, the MI355X (gfx950) implementation of the AV->LLM hot path of
 * rishabhjain16/audio-visual-llm `src/clip_whisper`.
 *
 * The reference has no FFI of its own: its seam is the Python class surface
 * (src/clip_whisper/models/clip_whisper_model.py, trainer/clip_whisper_trainer.py).  The entry points
 * below are what a ctypes binding under those classes calls; each cites the reference interface it
 * replaces.  Plain pointers and sizes only: every `const void*`/`void*` is a DEVICE pointer unless the
 * comment says host; `stream` is a hipStream_t passed as void* (NULL = default stream).  All functions
 * are asynchronous on `stream`, return 0 on success or an AVLLM_ERR_* code, and leave a message for
 * avllm_last_error() (thread local).  No function allocates device memory: scratch comes from the
 * caller-provided workspace.
 *
 * dtype: AVLLM_F32 (strict-parity mode, fp32 MFMA) or AVLLM_BF16 (bf16 storage, fp32 accumulate).
 */
#ifndef AVLLM_H
#define AVLLM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AVLLM_F32 0
#define AVLLM_BF16 1
#define AVLLM_ACT_NONE 0
#define AVLLM_ACT_GELU 1        /* erf GELU  (Whisper, HF:models/whisper/modeling_whisper.py:618-619,:405) */
#define AVLLM_ACT_QUICK_GELU 2  /* x*sigmoid(1.702x) (CLIP, HF:models/clip/modeling_clip.py:346-350) */
#define AVLLM_ACT_SILU 3
#define AVLLM_LORA_PAD 64       /* LoRA rank is stored padded to this many columns */

const char* avllm_last_error(void);
int avllm_version(void);

/* ---------------------------------------------------------------- op level -------------------- */

/* C[M,N] = act(alpha*(A.B^T + A2.B2^T) + bias) + R ; replaces every nn.Linear / F.linear on the path
 * (modality_connector.py:43-44; HF q/k/v/out/fc1/fc2/gate/up/down/lm_head) incl. the peft LoRA add. */
typedef struct avllm_gemm_desc {
    const void* A;  const void* B;      /* A [M,K] row stride lda; B [N,K] row stride ldb (nn.Linear weight) */
    const void* A2; const void* B2;     /* optional second K segment [M,K2] / [N,K2] (LoRA), NULL if K2==0 */
    void* C;                            /* [M,N] row stride ldc */
    const void* bias;                   /* [N] in `dtype`, or NULL */
    const void* R;                      /* residual [M,N] (or [r_mod,N]) in `dtype`, or NULL */
    int64_t lda, ldb, lda2, ldb2, ldc, ldr;
    int32_t M, N, K, K2;                /* K, K2 multiples of 64 */
    int32_t dtype;                      /* of A,B,A2,B2,bias,R (and C unless out_f32) */
    int32_t out_f32;                    /* store C as float even in bf16 mode */
    int32_t act;
    float alpha;
    int32_t r_mod;                      /* >0: residual row = m % r_mod (broadcast position embeddings) */
    int32_t g_in, g_out, g_off;         /* g_in>0: output row = (m/g_in)*g_out + g_off + m%g_in */
    uint32_t drop_seed;                 /* drop_p>0: the product (before +R) is multiplied by the dropout mask */
    float drop_p;                       /*   keep(drop_seed, m*N+n, p)/(1-p)  -- see avllm_dropout */
    uint32_t a_drop_seed;               /* a_drop_p>0: the A operand is dropout(A), mask index m*K+k (bf16, N==64 rank-side GEMM only): */
    float a_drop_p;                     /*   peft's lora_A(dropout(x)) without materialising dropout(x) */
    int32_t n_valid;                    /* N == 64 rank-side GEMM: only rows [0,n_valid) of B are non-zero (rank padded to 64); the other
                                         * output columns are written as zeros without being computed.  0 = all N */
    const uint32_t* seed_dev;           /* optional DEVICE word added to drop_seed / a_drop_seed at run time (see avllm_step_state) */
} avllm_gemm_desc;
int avllm_gemm(const avllm_gemm_desc* d, void* stream);
/* A/B testing only: force one bf16 tiling (0 = automatic choice; same values as env AVLLM_GEMM_VARIANT): 1 = 128x128, 2 = 256x128 ring, 5 / 6 = 256x256
 * with 16 waves (one-shot / persistent), 7 / 8 = 256x256 with 4 waves (one-shot / persistent: the default for every lean-epilogue call on a chip-filling
 * grid), 9 = 256x128 persistent with two workgroups per CU (csrc/gemm_dp.hip).  A variant that cannot take a call leaves it to the automatic choice. */
int avllm_set_gemm_variant(int v);
/* A/B testing only: the library's experiment switches live in ONE table that is filled once per process from the environment
 * (AVLLM_<NAME>) and changed afterwards only through this call.  Names: "DECODE_FUSED" (0 = general 10-launch decode layer),
 * "DEC_AL" (force an activation-load form of avllm_dec_proj: 2 | 4), "LORA_UNBATCHED" (1 = one launch per adapter), "F8_UNFUSED_QUANT"
 * (1 = separate quantiser passes), "F8_FAST" (0 = reference-grade fp8 GEMM), "ATTN_SHORT" (0 = general attention kernel for T <= 272),
 * "NARROW_EPILOGUE", "TN_CHUNK", "GEMM_GW" (tile-column group width of the persistent GEMM's walk on tall shapes: 0 = row-major walk,
 * n = groups of n columns), "GEMM_DBG" (only read by builds made with -DAVLLM_EXPERIMENT_KNOBS).  Unknown name -> error. */
int avllm_set_knob(const char* name, int32_t value);

/* out[I,J] (f32, row stride ldo) += alpha * sum_m P[m,i]*Q[m,j]; LoRA dA/dB (autograd of peft lora.Linear) */
int avllm_gemm_tn(const void* P, int64_t ldp, int32_t I, const void* Q, int64_t ldq, int32_t J, int32_t M,
                  float* out, int64_t ldo, float alpha, int32_t dtype, void* stream);
/* same, with dropout(seed,p) applied on the fly to the WIDE operand (mask index m*width+col; bf16 MFMA path only) */
int avllm_gemm_tn_drop(const void* P, int64_t ldp, int32_t I, const void* Q, int64_t ldq, int32_t J, int32_t M,
                       float* out, int64_t ldo, float alpha, uint32_t drop_seed, float drop_p, int32_t dtype, void* stream);

/* The q / k / v adapters of one decoder layer batched (bf16, rank <= 16, 1..3 adapters; peft lora.Linear on q_proj, k_proj, v_proj,
 * clip_whisper_model.py:961-1005).  Rank-side products C_j[M,64] = alpha * A_j[M,K_j] B_j[R,K_j]^T (columns >= 16 written as zeros):
 * shared != 0: every adapter reads A[0] through its own dropout mask (forward lora_A(dropout(x))); shared == 0: adapter j reads A[j]
 * (backward dt_j = s dy_j B_j). */
int avllm_lora_rank3(const void* const* A, const int64_t* lda, const int32_t* K, const void* const* B, const int64_t* ldb, void* const* C,
                     const int64_t* ldc, const uint32_t* seeds, int32_t nj, int32_t M, int32_t R, float alpha, float p,
                     const uint32_t* seed_dev, int32_t shared, int32_t dtype, void* stream);
/* Reductions over tokens for the same adapters (fp32 atomic accumulation into out_j, as avllm_gemm_tn):
 * shared != 0: out_j[R, NB] += alpha * Small_j^T . dropout_j(Big)            (dA_j; Big [M, NB] read once)
 * shared == 0: out_j[ncol_j, R] += alpha * Big[:, col0_j:+ncol_j]^T . Small_j (dB_j; the ranges tile Big's NB columns in 128-column units) */
int avllm_gemm_tn_multi(const void* Big, int64_t ldb, int32_t NB, const void* const* Small, const int64_t* lds, float* const* out,
                        const int64_t* ldo, const int32_t* col0, const int32_t* ncol, const uint32_t* seeds, int32_t nj, int32_t R,
                        int32_t M, float alpha, float p, const uint32_t* seed_dev, int32_t shared, int32_t dtype, void* stream);

/* Input gradient of up to three LoRA adapters that share one input x, under lora_dropout, in one pass (bf16):
 *   out[M,N] = R[M,N] + sum_j keep(seeds[j] (+ *seed_dev), m*N+n, p)/(1-p) * (T_j[M,r] . A_j[r,N])
 * autograd of peft's lora_B(lora_A(dropout(x))) w.r.t. x (clip_whisper_model.py:961-1005).  T_j [M, >= 32 cols] (zeros past the rank, row
 * stride ldt[j]); AT_j = the padded transposed image of A_j [N, >= 32 cols] (avllm_lora_pack's AT_pad, row stride ldat[j]); T / AT / ldt /
 * ldat / seeds are HOST arrays of nj <= 3 entries; out may alias R.  N % 128 == 0, r <= 32. */
int avllm_lora_dx_masked(const void* const* T, const int64_t* ldt, const void* const* AT, const int64_t* ldat, const uint32_t* seeds,
                         int32_t nj, int32_t r, const void* R, int64_t ldr, void* out, int64_t ldo, int32_t M, int32_t N, float p,
                         const uint32_t* seed_dev, int32_t dtype, void* stream);

/* ---- block-scaled fp8 (OCP MX: e4m3 elements + one E8M0 scale per 32 K elements; BASELINE config 5, "fp8 MFMA").  The reference has no
 * such mode (clip_whisper_model.py:164: only use_fp16); these entry points give nn.Linear's y = act(x W^T + b) + R on
 * v_mfma_scale_f32_16x16x128_f8f6f4 with fp32 accumulation, bf16 output.
 * avllm_mx_quantize: x [R,K] (bf16 or f32, row stride ldx) -> q uint8 [R,K] (row stride ldq bytes) + the scale image of
 * avllm_mx_scale_bytes(R,K) bytes.  layout 0 = activation side, 1 = weight side (csrc/fp8.hip, "Formats"). */
size_t avllm_mx_scale_bytes(int32_t R, int32_t K);
int avllm_mx_quantize(const void* x, int64_t ldx, int32_t R, int32_t K, void* q, int64_t ldq, void* scales, int32_t layout, int32_t dtype,
                      void* stream);
typedef struct avllm_gemm_f8_desc {
    const void* A;  const void* SA;     /* activations: q [M,K] (layout 0 scales) */
    const void* B;  const void* SB;     /* weights:     q [N,K] (layout 1 scales), nn.Linear orientation */
    void* C;                            /* bf16 [M,N] */
    const void* bias;                   /* bf16 [N] or NULL */
    const void* R;                      /* bf16 residual [M,N] or NULL (may alias C) */
    int64_t lda, ldb, ldc, ldr;
    int32_t M, N, K;                    /* K % 128 == 0 */
    int32_t act;
    /* Quantised output (the activation of the NEXT fp8 projection, e.g. fc1 -> fc2): when Cq != NULL the result act(A.B^T + bias) is
     * block-scaled to e4m3 in the epilogue, straight from the fp32 accumulators -- codes uint8 [M,N] (row stride ldcq bytes) + the layout-0
     * scale image of avllm_mx_scale_bytes(M,N) bytes in SCq -- and the bf16 C is NOT written (C may be NULL; R must be NULL; N % 32 == 0).
     * Only the persistent kernel implements it: AVLLM_ERR_UNSUPPORTED for calls it does not take (avllm_gemm_f8_takes_quantised_output). */
    void* Cq; void* SCq;
    int64_t ldcq;
} avllm_gemm_f8_desc;
int avllm_gemm_f8_takes_quantised_output(const avllm_gemm_f8_desc* d);
int avllm_gemm_f8(const avllm_gemm_f8_desc* d, void* stream);
/* nn.LayerNorm (b != NULL; HF whisper / clip encoder layers) or LlamaRMSNorm (b == NULL) of bf16 rows with the result block-scaled to e4m3 in
 * the same pass: q uint8 [rows, d] (row stride ldq) + the layout-0 scale image, ready as the A operand of avllm_gemm_f8.  y (bf16 copy of the
 * normalised rows) and rstd_out (RMSNorm's per-row 1/rms, saved for the backward) are optional.  d % 128 == 0, d <= 8192. */
int avllm_norm_mxq(const void* x, const void* w, const void* b, void* y, float* rstd_out, void* q, int64_t ldq, void* scales, int64_t rows,
                   int32_t d, float eps, void* stream);

/* Per-step scalars kept in DEVICE memory so that a training step is the same launch sequence every time and can be captured in a
 * hipGraph (trainer/clip_whisper_trainer.py:433-490 recomputes them on the host each step: scheduler.step() :464, the optimizer's
 * step count, torch's dropout RNG).  avllm_step_advance is a one-thread kernel: step += 1, then
 *   lr           = cosine schedule of _setup_optimizer :210-230 at (step-1): base_lr*(1+cos(pi*s/total))/2, or linear warm-up over
 *                  warmup_steps followed by the cosine over the remaining steps (transformers get_cosine_schedule_with_warmup)
 *   bc1, bc2_sqrt = 1-beta1^step, sqrt(1-beta2^step)                           (torch.optim.AdamW bias corrections)
 *   dropout_seed = (step+skipped)*0x9E3779B1 + rank*0x85EBCA6B + 12345         (this step's LoRA dropout mask seed)
 * Consumers: avllm_llama.dropout_seed_dev (-> &state->dropout_seed), avllm_adamw_step(state). */
typedef struct avllm_step_state {
    uint32_t step;                      /* optimizer steps started (1-based after the first advance) */
    uint32_t dropout_seed;
    float lr, bc1, bc2_sqrt;
    float skipped;                      /* steps whose update was skipped by the non-finite guard (see avllm_adamw_step) */
    uint32_t reserved[2];
} avllm_step_state;
typedef struct avllm_schedule {
    float base_lr, beta1, beta2;
    int32_t warmup_steps, total_steps;
    uint32_t rank;
} avllm_schedule;
int avllm_step_advance(avllm_step_state* state_dev, const avllm_schedule* sched, void* stream);

/* Sequence operators of the non-default connectors (modality_connector.py:111-380), activations token-major [B, T, C]:
 * im2col for nn.Conv1d(kernel_size 3, padding 1, stride 1 | 2): cols [B * Tout, 3 C], column kw * C + c = x[b, stride t' + kw - 1, c] (zero outside
 * the sequence), Tout = (T - 1) / stride + 1; the convolution is then avllm_gemm with the weight reshaped to [out, kw * C + c]. */
int avllm_im2col_k3(const void* x, void* cols, int32_t B, int32_t T, int32_t C, int32_t stride, int32_t dtype, void* stream);
/* nn.GroupNorm(groups, C) of the [B, C, T] view of x [B, T, C]: mean / variance over (T, C / groups) per (item, group), affine w, b [C],
 * then act (AVLLM_ACT_*).  (C / groups) % 8 == 0. */
int avllm_groupnorm_tokens(const void* x, const void* w, const void* b, void* y, int32_t B, int32_t T, int32_t C, int32_t groups,
                           float eps, int32_t act, int32_t dtype, void* stream);

/* nn.LayerNorm (HF whisper :379-413, clip :362-384) */
int avllm_layernorm(const void* x, const void* w, const void* b, void* y, int64_t rows, int32_t d, float eps,
                    int32_t dtype, void* stream);
/* LlamaRMSNorm fwd (HF:models/llama/modeling_llama.py:62-67); rstd [rows] optional */
int avllm_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int32_t d, float eps,
                      int32_t dtype, void* stream);
/* dx_out = dres_in + d(rmsnorm)/dx . dy   (dres_in may be NULL, dx_out may alias dres_in) */
int avllm_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres_in,
                      void* dx_out, int64_t rows, int32_t d, int32_t dtype, void* stream);
/* rotary embedding in place on [rows = B*T, heads, hd] slices (row stride ld elements); position = pos0 + row % T;
 * inverse=1 applies the transpose rotation (backward).  HF:models/llama/modeling_llama.py:129-160 */
int avllm_rope(void* x, int64_t ld, int64_t rows, int32_t T, int32_t heads, int32_t hd, int32_t pos0, float theta,
               int32_t inverse, int32_t dtype, void* stream);
/* h = silu(g)*u with gu = [g | u] ([M,2F]); HF:models/llama/modeling_llama.py:175 */
int avllm_swiglu_fwd(const void* gu, void* h, int64_t M, int32_t F, int32_t dtype, void* stream);
int avllm_swiglu_bwd(const void* dh, const void* gu, void* dgu, int64_t M, int32_t F, int32_t dtype, void* stream);
/* softmax(QK^T*scale [+causal]) V.  q/k/v/o: row = token (b*T+t), head h at column h*hd; row strides in elements.
 * lse [B,H,Tq] (natural log) optional.  impl 0 = MFMA flash kernel (bf16 only), 1 = reference-grade scalar kernel.
 * kv_heads (0 = H): grouped-query attention, k/v (and dk/dv) hold kv_heads heads and query head h reads head h/(H/kv_heads)
 * (HF repeat_kv, models/llama/modeling_llama.py:203-212).
 * HF eager_attention_forward: whisper :215-238, clip :259-277, llama sdpa path. */
int avllm_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int32_t B, int32_t Tq,
                        int32_t Tk, int32_t H, int32_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                        float scale, int32_t causal, int32_t dtype, int32_t impl, int32_t kv_heads, void* stream);
int avllm_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                        const float* lse, void* dq, void* dk, void* dv, float* delta_ws, int32_t B, int32_t T,
                        int32_t H, int32_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq,
                        int64_t lddk, int64_t lddv, float scale, int32_t causal, int32_t dtype, int32_t impl,
                        int32_t kv_heads, void* stream);
/* shifted causal-LM cross entropy (HF:loss/loss_utils.py:49-71): row (b,t) is scored against labels[b,t+1],
 * ignore_index -100 and the last position.  loss_sum/count are ACCUMULATED (zero them first). row_lse [B*T]. */
int avllm_ce_fwd(const void* logits, int64_t ld, const int64_t* labels, int32_t B, int32_t T, int32_t V,
                 float* row_lse, float* loss_sum, float* count, int32_t dtype, void* stream);
/* The same attention for short non-causal sequences (T <= 272, head_dim 64, bf16: the CLIP towers) with the OUTPUT block-scaled to e4m3 in the
 * kernel's epilogue: codes oq [B*T, ldoq] + the layout-0 scale image of an avllm_gemm_f8 A operand (avllm_mx_scale_bytes(B*T, H*hd) bytes),
 * bit-identical to avllm_mx_quantize of the bf16 output.  Used by the fp8 encoders: the bf16 attention output is never written. */
int avllm_attention_fwd_mxq(const void* q, const void* k, const void* v, void* oq, int64_t ldoq, void* scales, int32_t B, int32_t T, int32_t H,
                            int32_t hd, int64_t ldq, int64_t ldk, int64_t ldv, float scale, void* stream);
/* dlogits = (softmax - onehot) * (grad_scale / *count) for scored rows, 0 elsewhere (may alias logits) */
int avllm_ce_bwd(const void* logits, int64_t ld, const int64_t* labels, const float* row_lse, const float* count,
                 float grad_scale, void* dlogits, int32_t B, int32_t T, int32_t V, int32_t dtype, void* stream);
int avllm_argmax_rows(const void* logits, int64_t ld, int64_t rows, int32_t V, int64_t* out, int32_t dtype, void* stream);
/* out[i,:] = table[ids[i],:] ; llm.get_input_embeddings() (clip_whisper_model.py:464-487) */
int avllm_embedding(const void* table, const int64_t* ids, void* out, int64_t n, int32_t d, int32_t dtype, void* stream);
int avllm_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n, void* stream);
/* y = act(x) + r (r may be NULL; y may alias x): the activation / residual steps of DeepModalityConnector (modality_connector.py:91-108) */
int avllm_act_residual(const void* x, const void* r, void* y, int64_t n, int32_t act, int32_t dtype, void* stream);
/* nn.Dropout(p) of peft's lora.Linear (lora_dropout, clip_whisper_model.py:973-982): y = x * keep / (1-p) with the
 * counter-based mask keep(seed, row*d+col, p) (csrc/common.h av_keep); the same (seed,p) regenerates the same mask. */
int avllm_dropout(const void* x, void* y, int64_t rows, int32_t d, uint32_t seed, float p, int32_t dtype, void* stream);

/* Whisper conv stem im2col (HF:models/whisper/modeling_whisper.py:618-619).
 * conv1: mel f32 [B,80,T] -> cols [B*T, Kpad] with column c*3+kw = mel[b,c,t+kw-1] (zero padded, Kpad>=240)
 * conv2: h [B*T,d] -> cols [B*(T/2), 3d] with column kw*d+c = h[b,2t'+kw-1,c]                                */
int avllm_whisper_im2col1(const float* mel, void* cols, int32_t B, int32_t n_mels, int32_t T, int32_t Kpad,
                          int32_t dtype, void* stream);
int avllm_whisper_im2col2(const void* h, void* cols, int32_t B, int32_t T, int32_t d, int32_t dtype, void* stream);
/* CLIP patchify (HF:models/clip/modeling_clip.py:202-218): frames f32 [N,3,S,S] -> [N*(S/p)^2, Kpad>=3*p*p], column
 * c*p*p + ky*p + kx ; and the class-token rows  x[n,0,:] = class_emb + pos[0,:]                                */
int avllm_clip_patchify(const float* frames, void* cols, int32_t N, int32_t S, int32_t p, int32_t Kpad, int32_t dtype,
                        void* stream);
int avllm_clip_cls_rows(const void* class_emb, const void* pos, void* x, int32_t N, int32_t tokens, int32_t d,
                        int32_t dtype, void* stream);
/* encode()+forward() glue in one pass (clip_whisper_model.py:320-374,426-450,621-707):
 * virtual sequence X[b] = [prompt_emb[b,0:P] ; fs*a[b,t] + (1-fs)*v[b,t] (zero past each length), t<L]
 * (a or v may be NULL -> the other alone, scale 1).  out[b,0:S_out] = X (S_out==P+L), AdaptiveAvgPool1d
 * (P+L > S_out) or linear interpolation align_corners=True (P+L < S_out).                                  */
int avllm_fuse_pool(const void* a, int32_t Ta, const void* v, int32_t Tv, const void* prompt_emb, int32_t P,
                    void* out, int32_t B, int32_t L, int32_t S_out, int32_t D, float fusion_scale, int32_t dtype,
                    void* stream);
/* clip_grad_norm_ + AdamW (trainer/clip_whisper_trainer.py:457-464,171-232) on a flat fp32 buffer, no host sync:
 * sumsq accumulates sum(g^2); the step reads it, clips with coef=min(1,max_norm/(sqrt(sumsq)+1e-6)).
 * Non-finite guard (trainer :444-452 skips backward + optimizer on a NaN/Inf loss): when *sumsq or *guard (e.g. the step's
 * loss_sum; may be NULL) is not finite the launch leaves p, m, v untouched and adds 1 to *skipped (device float, may be NULL). */
int avllm_grad_sumsq(const float* g, int64_t n, float* sumsq, void* stream);
/* The same sum in a fixed order (no float atomics): *sumsq = sum g^2, overwritten.  partials = scratch of nparts floats (<= 1024 are used).
 * Bit-reproducible, so data-parallel replicas that clip identical all-reduced gradients keep bit-identical parameters. */
int avllm_grad_sumsq_det(const float* g, int64_t n, float* partials, int32_t nparts, float* sumsq, void* stream);
int avllm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int32_t step, const float* sumsq, float max_norm,
                     float grad_prescale, const float* guard, float* skipped, const avllm_step_state* state_dev, void* stream);
/* state_dev != NULL: lr and the bias corrections come from *state_dev (the `lr` and `step` arguments are ignored); a skipped step is
 * also counted in state_dev->skipped and takes state_dev->step back by one, so that neither the schedule nor Adam's bias corrections
 * advance on it (the reference skips optimizer.step() AND scheduler.step(), trainer/clip_whisper_trainer.py:444-452). */
/* build the four padded operand images of one LoRA pair from the fp32 masters A [r,din], B [dout,r]:
 * A_pad [64,din], AT_pad [din,64] (row stride ld_at), B_pad [dout,64], BT_pad [64,dout] in `dtype` */
int avllm_lora_pack(const float* A, const float* Bm, int32_t r, int32_t din, int32_t dout, void* A_pad, void* AT_pad,
                    int64_t ld_at, void* B_pad, void* BT_pad, int32_t dtype, void* stream);
/* the same for every adapter of a model in one launch: `items_dev` is a table of n rows in DEVICE memory (built once).
 * Only the r real rank rows/columns are written: the images must be zero-initialised once (their padding never changes). */
typedef struct avllm_lora_pack_item {
    const float *A, *B;              /* fp32 masters: A [r,din], B [dout,r] */
    void *A_pad, *AT_pad, *B_pad, *BT_pad;
    int64_t ld_at, dout;
} avllm_lora_pack_item;
int avllm_lora_pack_batch(const avllm_lora_pack_item* items_dev, int32_t n, int32_t r, int32_t din, int32_t dtype, void* stream);

/* ---------------------------------------------------------------- input features on device ----------------
 * The reference builds the hot path's inputs on CPU workers, one sample at a time (src/clip_whisper/data/simple_dataset.py):
 *   audio :156-186  whisper_processor(audio, sampling_rate=16000).input_features then F.layer_norm(features, features.shape)
 *   video :191-264  clip_processor(images=frame)["pixel_values"] per RGB uint8 frame
 * These entry points do the same arithmetic on the GPU from the raw samples / raw uint8 frames (4x fewer PCIe bytes for
 * video than fp32 pixel_values).  Tables/plans are small device buffers the caller allocates and initialises once
 * (synchronous copy); the compute calls only enqueue kernels. */

/* Whisper log-mel: wave f32 [B, n] (row stride ld, mono 16 kHz, rows zero-padded to n; n > 480000 is truncated) ->
 * out f32 [B,80,3000]: 30 s zero pad, reflect pad 200, Hann(400) hop 160, |DFT|^2 (float64), slaney mel 80, log10, clamp to
 * max-8, (x+4)/4 (WhisperFeatureExtractor, feature_extraction_whisper.py:105-133); normalize != 0 adds the dataset's
 * whole-tensor layer norm (simple_dataset.py:181-183). */
size_t avllm_logmel_table_bytes(void);
int avllm_logmel_table_init(void* table_dev, int32_t n_mels);      /* n_mels: 80 (Whisper tiny..large-v2) or 128 (large-v3: feature_size=128) */
size_t avllm_logmel_workspace_bytes(int32_t B, int32_t n_mels);
int avllm_logmel(const void* table, const float* wave, int32_t B, int32_t n, int64_t ld, int32_t normalize, int32_t n_mels, float* out,
                 void* ws, size_t ws_bytes, void* stream);      /* out f32 [B, n_mels, 3000]; n_mels must be the table's */

/* CLIPImageProcessor on device: frames u8 [N,H,W,3] RGB -> out [N,3,image,image] (dtype f32 or bf16): resize so the shorter
 * edge is `image` (Pillow BICUBIC, bit-exact 8-bit fixed point: horizontal pass, uint8, vertical pass), centre crop, x/255,
 * (x-mean)/std.  A plan is specific to (H, W, image, mean, std). */
size_t avllm_clip_preproc_plan_bytes(int32_t H, int32_t W, int32_t image);
int avllm_clip_preproc_plan_init(void* plan_dev, int32_t H, int32_t W, int32_t image, const float* mean3, const float* std3);
size_t avllm_clip_preproc_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t image);
int avllm_clip_preproc(const void* plan, const uint8_t* frames, int32_t N, int32_t H, int32_t W, int32_t image, void* out,
                       int32_t dtype, void* ws, size_t ws_bytes, void* stream);

/* Live kernel timing for bench.py: between begin and end every avllm_gemm launch (direct or inside the model-level
 * calls) is bracketed by two HIP events on its stream.  out[0]=sum of GEMM ms, out[1]=sum of algorithmic FLOPs
 * (2*M*N*(K+K2)), out[2]=launches timed. */
int avllm_profile_begin(int32_t max_launches);
int avllm_profile_enable(int32_t on);      /* pause (0) / resume (1) bracketing between begin and end; recorded launches are kept */
int avllm_profile_end(double* out);

/* ---------------------------------------------------------------- model level ------------------ */

/* one pre-LN transformer encoder block with biases (Whisper encoder layer / CLIP encoder layer) */
typedef struct avllm_enc_layer {
    const void *ln1_w, *ln1_b;
    const void *wqkv, *bqkv;     /* [3d,d], [3d]  rows = [q;k;v]; whisper k bias = 0 */
    const void *wo, *bo;
    const void *ln2_w, *ln2_b;
    const void *w1, *b1, *w2, *b2;
    /* fp8 mode (avllm_whisper.fp8 / avllm_clip.fp8 != 0): the four weight matrices once more as block-scaled fp8 (avllm_mx_quantize layout 1):
     * codes [rows, K] and scale images.  The bf16 matrices above are then unused by the encoder forward. */
    const void *wqkv8, *sqkv8, *wo8, *so8, *w18, *s18, *w28, *s28;
} avllm_enc_layer;

typedef struct avllm_whisper {
    int32_t dtype, d, heads, layers, ffn, n_mels, n_ctx, k1pad;
    const void *conv1_w, *conv1_b;   /* [d,k1pad] (col c*3+kw, zero padded), [d] */
    const void *conv2_w, *conv2_b;   /* [d,3d] (col kw*d+c), [d] */
    const void* pos;                 /* [n_ctx,d] */
    const avllm_enc_layer* layer;    /* host array [layers] */
    const void *lnf_w, *lnf_b;
    int32_t fp8;                     /* 1: layer projections run on the block-scaled fp8 matrix pipe (dtype must be AVLLM_BF16) */
} avllm_whisper;
size_t avllm_whisper_workspace_bytes(const avllm_whisper* w, int32_t B);
/* ClipWhisperModel.encode_audio minus the connector (clip_whisper_model.py:1067-1104 ->
 * WhisperEncoder.forward): mel f32 [B,80,2*n_ctx] -> out [B,n_ctx,d] */
int avllm_whisper_encoder_fwd(const avllm_whisper* w, const float* mel, int32_t B, void* out, void* ws,
                              size_t ws_bytes, void* stream);

typedef struct avllm_clip {
    int32_t dtype, d, heads, layers, ffn, image, patch, tokens;
    float eps;
    const void* patch_w;             /* [d, 3*p*p] */
    const void *class_emb, *pos;     /* [d], [tokens,d] */
    const void *pre_ln_w, *pre_ln_b;
    const avllm_enc_layer* layer;    /* host array [layers] */
    int32_t fp8;                     /* as avllm_whisper.fp8 */
    int32_t frames_bf16;             /* 1: `frames` of avllm_clip_vision_cls_fwd are bf16 (what avllm_clip_preproc writes with dtype bf16: half the bytes
                                      * of the reference's fp32 pixel_values on the way into the patch embedding) */
} avllm_clip;
size_t avllm_clip_workspace_bytes(const avllm_clip* c, int32_t N);
/* ClipWhisperModel.encode_video minus the connector (clip_whisper_model.py:1108-1142 -> CLIPVisionModel.forward,
 * last_hidden_state[:,0], no post_layernorm): frames f32 [N,3,S,S] -> cls [N,d] */
int avllm_clip_vision_cls_fwd(const avllm_clip* c, const void* frames, int32_t N, void* cls, void* ws,
                              size_t ws_bytes, void* stream);

typedef struct avllm_lora_mod {      /* padded operand images (see avllm_lora_pack); NULL = no adapter */
    const void *A_pad, *AT_pad, *B_pad, *BT_pad;
    int64_t ld_at;                   /* row stride of AT_pad: 64, or 192 when q/k/v share one [din,192] image */
    float *gA, *gB;                  /* fp32 grads [r,din], [dout,r] (accumulated) */
} avllm_lora_mod;

typedef struct avllm_llama_layer {
    const void *ln1_w, *ln2_w;
    const void *wqkv;                /* [d+2*dkv, d] rows [q;k;v]  (dkv = kv_heads*hd; 3d without GQA) */
    const void *wo;                  /* [d,d] */
    const void *wgu;                 /* [2f,d] rows [gate;up] */
    const void *wdown;               /* [d,f] */
    /* transposed images for dX = dY.W (training only; NULL for inference): */
    const void *wqkv_t;              /* [d, d+2*dkv] */
    const void *wo_t;                /* [d,d] */
    const void *wgu_t;               /* [d,2f] */
    const void *wdown_t;             /* [f,d] */
    avllm_lora_mod lora[4];          /* q,k,v,o */
    /* fp8 mode (avllm_llama.fp8 != 0): forward-pass images of the four frozen matrices (avllm_mx_quantize layout 1).  The backward pass keeps
     * using the bf16 transposed images: gradients are not quantised. */
    const void *wqkv8, *sqkv8, *wo8, *so8, *wgu8, *sgu8, *wdown8, *sdown8;
} avllm_llama_layer;

typedef struct avllm_llama {
    int32_t dtype, d, heads, layers, ffn, vocab, lora_r;
    int32_t kv_heads;                /* grouped-query attention: key/value heads (0 = heads).  wqkv is [(heads+2*kv_heads)*hd, d],
                                      * the k/v adapters' B is [kv_heads*hd, r], the KV cache rows are kv_heads*hd wide */
    float eps, theta, lora_scale;
    float lora_dropout;              /* applied by avllm_llama_lora_fwd_loss/_bwd only (training); 0 = off */
    uint32_t dropout_seed;           /* module j of layer l uses seed dropout_seed + 4*l + j; change it every step */
    const uint32_t* dropout_seed_dev;/* optional DEVICE word added to dropout_seed at run time (graph-replayable steps: avllm_step_state) */
    const void* embed;               /* [vocab,d] */
    const void* norm_w;
    const void* lm_head;             /* [vocab,d] */
    const void* lm_head_t;           /* [d,vocab] (training) */
    const avllm_llama_layer* layer;  /* host array [layers] */
    int32_t fp8;                     /* 1: the frozen projections of avllm_llama_lora_fwd_loss (q/k/v/o base terms, gate/up, down, lm_head) run on
                                      * the block-scaled fp8 matrix pipe (BASELINE config 5); LoRA terms, attention, norms and the whole
                                      * backward pass stay bf16 */
    const void *lm_head8, *slm_head8;
    /* rope_orig_ctx > 0: "llama3" RoPE frequency scaling of Llama-3.1 / 3.2 checkpoints (config.json rope_scaling: factor, low_freq_factor,
     * high_freq_factor, original_max_position_embeddings; HF:modeling_rope_utils.py _compute_llama3_parameters).  0 = plain RoPE. */
    float rope_factor, rope_low_freq_factor, rope_high_freq_factor;
    int32_t rope_orig_ctx;
} avllm_llama;

size_t avllm_llama_train_workspace_bytes(const avllm_llama* m, int32_t B, int32_t S);
/* self.llm(inputs_embeds, attention_mask=ones, labels) (clip_whisper_model.py:602-613 ->
 * LlamaForCausalLM.forward HF:models/llama/modeling_llama.py:435-488 + ForCausalLMLoss).
 * x [B,S,d]; labels int64 [B,S] with -100 already applied; logits [B,S,vocab] written when non-NULL
 * (always materialised internally in this version).  loss_sum/count accumulate (zero first); activations for
 * the backward pass stay in `ws`. */
int avllm_llama_lora_fwd_loss(const avllm_llama* m, const void* x, const int64_t* labels, int32_t B, int32_t S,
                              void* logits, float* loss_sum, float* count, void* ws, size_t ws_bytes, void* stream);
/* loss.backward() for the LoRA tensors only (frozen base weights => dX GEMMs + LoRA dA/dB).
 * Must follow avllm_llama_lora_fwd_loss with the same ws.  Gradient of  grad_scale * loss_sum / *count.
 * `after_layer` (may be NULL) is called on the host right after layer i's kernels are enqueued (31 -> 0) so a
 * data-parallel caller can launch that layer's gradient all-reduce on a side stream. */
typedef void (*avllm_layer_cb)(int32_t layer, void* user);
int avllm_llama_lora_bwd(const avllm_llama* m, const int64_t* labels, int32_t B, int32_t S, const float* count,
                         float grad_scale, void* ws, size_t ws_bytes, avllm_layer_cb after_layer, void* user,
                         void* stream);
/* The same backward pass in pieces: decoder layers layer_hi, layer_hi-1, ..., layer_lo (inclusive).  The piece that starts at the
 * last layer (layer_hi == layers-1) also runs the loss / lm_head / final-norm part; consecutive pieces must be called in descending
 * order on the same ws (the running residual gradient stays there).  Lets a data-parallel caller replay each piece from its own
 * hipGraph and launch that piece's gradient all-reduce while the next piece runs. */
int avllm_llama_lora_bwd_layers(const avllm_llama* m, const int64_t* labels, int32_t B, int32_t S, const float* count,
                                float grad_scale, void* ws, size_t ws_bytes, int32_t layer_hi, int32_t layer_lo,
                                avllm_layer_cb after_layer, void* user, void* stream);

/* Inference: prefill on inputs_embeds and single-token steps with a KV cache (llm.generate,
 * clip_whisper_model.py:1337-1340 -> GenerationMixin greedy).  kcache/vcache [layers][B][Tmax][d].
 * prefill writes positions [0,S) and returns the hidden state of the LAST position after the final norm
 * -> logits_last [B,vocab] f32; decode_step embeds `ids` [B], runs position `pos`, returns logits [B,vocab]. */
size_t avllm_llama_infer_workspace_bytes(const avllm_llama* m, int32_t B, int32_t S);
int avllm_llama_prefill(const avllm_llama* m, const void* x, int32_t B, int32_t S, void* kcache, void* vcache,
                        int32_t Tmax, float* logits_last, void* all_logits, void* ws, size_t ws_bytes, void* stream);
int avllm_llama_decode_step(const avllm_llama* m, const int64_t* ids, int32_t B, int32_t pos, void* kcache,
                            void* vcache, int32_t Tmax, float* logits, void* ws, size_t ws_bytes, void* stream);
/* The same step with the position in DEVICE memory: the token runs at position pos + *pos_dev (pos_dev may be NULL), so a captured
 * hipGraph of the step can be replayed for every token (advance *pos_dev between replays, e.g. with avllm_step_advance's sibling
 * avllm_pos_advance).  The caller guarantees pos + *pos_dev < Tmax. */
int avllm_llama_decode_step_at(const avllm_llama* m, const int64_t* ids, int32_t B, int32_t pos, const int32_t* pos_dev, void* kcache,
                               void* vcache, int32_t Tmax, float* logits, void* ws, size_t ws_bytes, void* stream);
/* 1 when a token step of B sequences on this model takes the fused bf16 path (one launch per projection + one attention launch per
 * layer), 0 when it takes the general path: tests and benchmarks assert which one they measured. */
int avllm_llama_decode_is_fused(const avllm_llama* m, int32_t B);
int avllm_pos_advance(int32_t* pos_dev, int32_t by, void* stream);

/* One projection of a decode token step (bf16, 1 <= M <= 16 rows, K % 128 == 0): C = epilogue(rmsnorm?(A) . W^T).  Every weight row
 * is streamed from HBM once; the fused forms remove the launches between the projections of LlamaDecoderLayer.forward:
 *   mode 0  plain: C[M,N] (+ R), bf16 or f32 out                                    (o_proj / down_proj + residual, lm_head)
 *   mode 1  SwiGLU: W = [gate; up] rows [2N, K]; C[M,N] = silu(A.gate^T) * (A.up^T)  (LlamaMLP.forward's act_fn(gate) * up)
 *   mode 2  q|k|v: W rows [dq + 2 dkv, K]; rotary embedding on q and k (pairs (i, i + hd/2), table rope[hd/2][2] = cos,sin of the
 *           position); q -> C[M,dq]; k, v -> cache rows kc/vc[m][pos + *pos_dev][dkv]  (apply_rotary_pos_emb + DynamicCache.update)
 * norm_w != NULL folds the preceding RMSNorm in: x * rsqrt(mean(x^2) + eps) * norm_w is applied to A on the fly. */
typedef struct avllm_dec_proj_desc {
    const void* A; int64_t lda;
    const void* W; int64_t ldw;
    const void* norm_w; float eps;
    int32_t M, K, N, mode;
    void* C; int64_t ldc; int32_t out_f32;
    const void* R; int64_t ldr;
    int32_t dq, dkv, hd;
    const float* rope;
    void *kc, *vc;
    int32_t Tmax, pos;
    const int32_t* pos_dev;
    /* LoRA side term of peft lora.Linear, + lora_scale * B (A x), added in the epilogue (before RoPE): lora_t [M, ld_lora_t] f32 holds the
     * rank-side products of this projection's (normalised) input, module j in columns [64 j, 64 j + lora_r) -- itself a mode-0 launch over
     * the A images; lora_b[j] = padded B image [rows of module j, 64] (avllm_lora_pack).  mode 2: j = q, k, v; mode 0: j = 0; not with mode 1.
     * lora_t == NULL: no adapters. */
    const float* lora_t; int64_t ld_lora_t;
    const void* lora_b[3];
    int32_t lora_r; float lora_scale;
} avllm_dec_proj_desc;
int avllm_dec_proj(const avllm_dec_proj_desc* d, void* stream);
/* Single-query attention over the cache rows [0, Tk + *tk_dev) (tk_dev may be NULL) of kc/vc [B][Tmax][(H/kv_group)*hd]: one pass with
 * an online softmax; q [B, H*hd] (row stride ldq), o [B, H*hd] (ldo).  LlamaAttention.forward with q_len == 1 (eager softmax(QK^T/sqrt(hd))V). */
int avllm_attention_decode(const void* q, int64_t ldq, const void* kc, const void* vc, void* o, int64_t ldo, int32_t B, int32_t H, int32_t hd,
                           int32_t Tk, const int32_t* tk_dev, int32_t Tmax, float scale, int32_t kv_group, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
