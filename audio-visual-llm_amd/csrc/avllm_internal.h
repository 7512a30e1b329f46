// Internal launchers (typed stream) behind the C ABI of include/avllm.h.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/avllm.h"

int av_gemm(const avllm_gemm_desc* d, hipStream_t st);
int av_gemm_tn(const void* P, long ldp, int I, const void* Q, long ldq, int J, int M, float* out, long ldo,
               float alpha, int dtype, hipStream_t st, uint32_t drop_seed = 0, float drop_p = 0.f, const uint32_t* seed_dev = nullptr);
int av_layernorm(const void* x, const void* w, const void* b, void* y, long rows, int d, float eps, int dtype, hipStream_t st);
int av_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, long rows, int d, float eps, int dtype, hipStream_t st);
int av_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres_in, void* dx_out,
                   long rows, int d, int dtype, hipStream_t st);
int av_rope(void* x, long ld, long rows, int T, int heads, int hd, int pos0, float theta, int inverse, int dtype, hipStream_t st);
int av_swiglu_fwd(const void* gu, void* h, long M, int F, int dtype, hipStream_t st);
int av_swiglu_bwd(const void* dh, const void* gu, void* dgu, long M, int F, int dtype, hipStream_t st);
bool av_attention_fwd_mxq_ok(int B, int T, int H, int hd, int dtype, int kv_heads);
int av_attention_fwd_mxq(const void* q, const void* k, const void* v, void* oq, long ldoq, void* osc, int B, int T, int H, int hd, long ldq, long ldk,
                         long ldv, float scale, hipStream_t st);
int av_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int Tq, int Tk, int H,
                     int hd, long ldq, long ldk, long ldv, long ldo, float scale, int causal, int dtype, int impl,
                     hipStream_t st, int kv_heads = 0);
int av_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                     void* dq, void* dk, void* dv, float* delta_ws, int B, int T, int H, int hd, long ldq, long ldk,
                     long ldv, long ldo, long lddq, long lddk, long lddv, float scale, int causal, int dtype, int impl,
                     hipStream_t st, int kv_heads = 0, const float* rope_tab = nullptr);
// rope_tab != NULL: dq and dk come out as gradients w.r.t. the PRE-RoPE q/k (inverse rotation applied in the kernels' epilogues)
bool av_attention_bwd_fuses_rope(int dtype, int hd, int impl);
int av_ce_fwd(const void* logits, long ld, const int64_t* labels, int B, int T, int V, float* row_lse, float* loss_sum,
              float* count, int dtype, hipStream_t st);
int av_ce_bwd(const void* logits, long ld, const int64_t* labels, const float* row_lse, const float* count,
              float grad_scale, void* dlogits, int B, int T, int V, int dtype, hipStream_t st);
int av_argmax_rows(const void* logits, long ld, long rows, int V, int64_t* out, int dtype, hipStream_t st);
int av_embedding(const void* table, const int64_t* ids, void* out, long n, int d, int dtype, hipStream_t st);
int av_cast(const void* src, int sdt, void* dst, int ddt, long n, hipStream_t st);
int av_whisper_im2col1(const float* mel, void* cols, int B, int n_mels, int T, int Kpad, int dtype, hipStream_t st);
int av_whisper_im2col2(const void* h, void* cols, int B, int T, int d, int dtype, hipStream_t st);
int av_clip_patchify(const void* frames, void* cols, int N, int S, int p, int Kpad, int dtype, hipStream_t st, int in_dtype = AV_F32);
int av_clip_cls_rows(const void* class_emb, const void* pos, void* x, int N, int tokens, int d, int dtype, hipStream_t st);
int av_fuse_pool(const void* a, int Ta, const void* v, int Tv, const void* prompt_emb, int P, void* out, int B, int L,
                 int S_out, int D, float fs, int dtype, hipStream_t st);
int av_grad_sumsq(const float* g, long n, float* sumsq, hipStream_t st);
int av_grad_sumsq_det(const float* g, long n, float* partials, int nparts, float* sumsq, hipStream_t st);
int av_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                  float wd, int step, const float* sumsq, float max_norm, float grad_prescale, const float* guard, float* skipped,
                  const avllm_step_state* state, hipStream_t st);
int av_lora_pack(const float* A, const float* Bm, int r, int din, int dout, void* A_pad, void* AT_pad, long ld_at,
                 void* B_pad, void* BT_pad, int dtype, hipStream_t st);
int av_kv_append(const void* k, const void* v, long ld, void* kc, void* vc, int B, int T, int pos0, int Tmax, int d,
                 int dtype, hipStream_t st);
int av_attention_decode(const void* q, long ldq, const void* kc, const void* vc, void* o, long ldo, int B, int H, int hd,
                        int Tk, int Tmax, float scale, int dtype, hipStream_t st, int G = 1);
int av_attention_decode1(const void* q, long ldq, const void* kc, const void* vc, void* o, long ldo, int B, int H, int hd, int Tk, const int* tk_dev,
                         int Tmax, float scale, int dtype, hipStream_t st, int G);
int av_norm_mxq(const void* x, const void* w, const void* b, void* y, float* rstd_out, void* q, long ldq, void* scales, long rows, int d, float eps,
                hipStream_t st);
bool av_lora_batch_supported(int dtype, int R, int nj);
int av_lora_rank3(const void* const* A, const long* lda, const int* K, const void* const* B, const long* ldb, void* const* C, const long* ldc,
                  const uint32_t* seeds, int nj, int M, int R, float alpha, float p, const uint32_t* seed_dev, int shared, int dtype, hipStream_t st);
int av_gemm_tn_multi(const void* Big, long ldb, int NB, const void* const* Small, const long* lds, float* const* out, const long* ldo,
                     const int* col0, const int* ncol, const uint32_t* seeds, int nj, int R, int M, float alpha, float p,
                     const uint32_t* seed_dev, int shared, int dtype, hipStream_t st);
bool av_dec_proj_supported(int dtype, int M, int K, int N, int mode, int hd);
int av_dec_proj(const avllm_dec_proj_desc* d, hipStream_t st);
struct AvRopeScale { float factor = 1.f, low_freq_factor = 1.f, high_freq_factor = 4.f; int orig_ctx = 0; };      // orig_ctx == 0: plain RoPE
int av_rope_table(float* tab, int T, int hd, int pos0, float theta, hipStream_t st, const int* pos_dev = nullptr, AvRopeScale sc = AvRopeScale());
int av_rope_tab(void* x, long ld, long rows, int T, int heads, int hd, const float* tab, int inverse, int dtype, hipStream_t st);
int av_dropout(const void* x, void* y, long rows, int d, uint32_t seed, float p, int dtype, hipStream_t st, const uint32_t* seed_dev = nullptr);
bool av_lora_dx_masked_supported(int dtype, int N, int r, const long* ldt, const long* ldat, int nj, long ldr, long ldo);
int av_lora_dx_masked(const void* const* T, const long* ldt, const void* const* AT, const long* ldat, const uint32_t* seeds, int nj, int r,
                      const void* R, long ldr, void* out, long ldo, int M, int N, float p, const uint32_t* seed_dev, int dtype, hipStream_t st);
int av_mx_quantize(const void* x, long ldx, int R, int K, void* q, long ldq, void* scales, int layout, int dtype, hipStream_t st);
int av_gemm_f8(const avllm_gemm_f8_desc* d, hipStream_t st);
int av_act_residual(const void* x, const void* r, void* y, long n, int act, int dtype, hipStream_t st);
int av_step_advance(avllm_step_state* state, const avllm_schedule* sched, hipStream_t st);
