// extern "C" surface of libavllm.so (include/avllm.h): error string + thin wrappers that turn the void*
// stream into a hipStream_t.  No torch types, no allocation.
#include "common.h"
#include "avllm_internal.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

int av_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- experiment knobs (common.h AvKnob)
#include <mutex>
#include <string.h>
static const struct { const char* name; int dflt; } g_knob_def[AV_KNOB_COUNT] = {
    {"DECODE_FUSED", 1}, {"DEC_AL", 0}, {"LORA_UNBATCHED", 0}, {"F8_UNFUSED_QUANT", 0}, {"F8_FAST", 1}, {"ATTN_SHORT", 1},
    {"NARROW_EPILOGUE", 0}, {"TN_CHUNK", 0}, {"GEMM_DBG", 0}, {"GEMM_GW", 0}};
static int g_knob[AV_KNOB_COUNT];
static std::once_flag g_knob_once;
static void knob_init() {
    for (int i = 0; i < AV_KNOB_COUNT; ++i) {
        char env[64];
        snprintf(env, sizeof(env), "AVLLM_%s", g_knob_def[i].name);
        const char* e = getenv(env);
        g_knob[i] = e ? atoi(e) : g_knob_def[i].dflt;
    }
}
int av_knob(int id) {
    std::call_once(g_knob_once, knob_init);
    return g_knob[id];
}

#define ST ((hipStream_t)stream)
extern "C" {
int avllm_set_knob(const char* name, int32_t value) {
    AV_CHECK_ARG(name, "set_knob: null name");
    std::call_once(g_knob_once, knob_init);
    for (int i = 0; i < AV_KNOB_COUNT; ++i)
        if (!strcmp(name, g_knob_def[i].name)) { g_knob[i] = value; return AV_OK; }
    return av_set_error(AV_ERR_ARG, "set_knob: unknown knob '%s'", name);
}
const char* avllm_last_error(void) { return g_err; }
int avllm_version(void) { return 100; }

int avllm_gemm(const avllm_gemm_desc* d, void* stream) { return av_gemm(d, ST); }
int avllm_gemm_tn(const void* P, int64_t ldp, int32_t I, const void* Q, int64_t ldq, int32_t J, int32_t M, float* out,
                  int64_t ldo, float alpha, int32_t dtype, void* stream) { return av_gemm_tn(P, ldp, I, Q, ldq, J, M, out, ldo, alpha, dtype, ST); }
int avllm_gemm_tn_drop(const void* P, int64_t ldp, int32_t I, const void* Q, int64_t ldq, int32_t J, int32_t M, float* out,
                       int64_t ldo, float alpha, uint32_t seed, float p, int32_t dtype, void* stream) { return av_gemm_tn(P, ldp, I, Q, ldq, J, M, out, ldo, alpha, dtype, ST, seed, p); }
int avllm_layernorm(const void* x, const void* w, const void* b, void* y, int64_t rows, int32_t d, float eps, int32_t dtype,
                    void* stream) { return av_layernorm(x, w, b, y, rows, d, eps, dtype, ST); }
int avllm_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int32_t d, float eps, int32_t dtype,
                      void* stream) { return av_rmsnorm_fwd(x, w, y, rstd, rows, d, eps, dtype, ST); }
int avllm_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres_in, void* dx_out,
                      int64_t rows, int32_t d, int32_t dtype, void* stream) { return av_rmsnorm_bwd(dy, x, w, rstd, dres_in, dx_out, rows, d, dtype, ST); }
int avllm_rope(void* x, int64_t ld, int64_t rows, int32_t T, int32_t heads, int32_t hd, int32_t pos0, float theta,
               int32_t inverse, int32_t dtype, void* stream) { return av_rope(x, ld, rows, T, heads, hd, pos0, theta, inverse, dtype, ST); }
int avllm_swiglu_fwd(const void* gu, void* h, int64_t M, int32_t F, int32_t dtype, void* stream) { return av_swiglu_fwd(gu, h, M, F, dtype, ST); }
int avllm_swiglu_bwd(const void* dh, const void* gu, void* dgu, int64_t M, int32_t F, int32_t dtype, void* stream) { return av_swiglu_bwd(dh, gu, dgu, M, F, dtype, ST); }
int avllm_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int32_t B, int32_t Tq, int32_t Tk,
                        int32_t H, int32_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, int32_t causal,
                        int32_t dtype, int32_t impl, int32_t kv_heads, void* stream) {
    return av_attention_fwd(q, k, v, o, lse, B, Tq, Tk, H, hd, ldq, ldk, ldv, ldo, scale, causal, dtype, impl, ST, kv_heads);
}
int avllm_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                        void* dk, void* dv, float* delta_ws, int32_t B, int32_t T, int32_t H, int32_t hd, int64_t ldq, int64_t ldk,
                        int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, float scale, int32_t causal,
                        int32_t dtype, int32_t impl, int32_t kv_heads, void* stream) {
    return av_attention_bwd(q, k, v, o, dout, lse, dq, dk, dv, delta_ws, B, T, H, hd, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, causal, dtype, impl, ST, kv_heads);
}
int avllm_ce_fwd(const void* logits, int64_t ld, const int64_t* labels, int32_t B, int32_t T, int32_t V, float* row_lse,
                 float* loss_sum, float* count, int32_t dtype, void* stream) { return av_ce_fwd(logits, ld, labels, B, T, V, row_lse, loss_sum, count, dtype, ST); }
int avllm_ce_bwd(const void* logits, int64_t ld, const int64_t* labels, const float* row_lse, const float* count, float grad_scale,
                 void* dlogits, int32_t B, int32_t T, int32_t V, int32_t dtype, void* stream) { return av_ce_bwd(logits, ld, labels, row_lse, count, grad_scale, dlogits, B, T, V, dtype, ST); }
int avllm_argmax_rows(const void* logits, int64_t ld, int64_t rows, int32_t V, int64_t* out, int32_t dtype, void* stream) { return av_argmax_rows(logits, ld, rows, V, out, dtype, ST); }
int avllm_embedding(const void* table, const int64_t* ids, void* out, int64_t n, int32_t d, int32_t dtype, void* stream) { return av_embedding(table, ids, out, n, d, dtype, ST); }
int avllm_cast(const void* src, int32_t sdt, void* dst, int32_t ddt, int64_t n, void* stream) { return av_cast(src, sdt, dst, ddt, n, ST); }
int avllm_dropout(const void* x, void* y, int64_t rows, int32_t d, uint32_t seed, float p, int32_t dtype, void* stream) { return av_dropout(x, y, rows, d, seed, p, dtype, ST); }
int avllm_whisper_im2col1(const float* mel, void* cols, int32_t B, int32_t n_mels, int32_t T, int32_t Kpad, int32_t dtype, void* stream) { return av_whisper_im2col1(mel, cols, B, n_mels, T, Kpad, dtype, ST); }
int avllm_whisper_im2col2(const void* h, void* cols, int32_t B, int32_t T, int32_t d, int32_t dtype, void* stream) { return av_whisper_im2col2(h, cols, B, T, d, dtype, ST); }
int avllm_clip_patchify(const float* frames, void* cols, int32_t N, int32_t S, int32_t p, int32_t Kpad, int32_t dtype, void* stream) { return av_clip_patchify(frames, cols, N, S, p, Kpad, dtype, ST); }
int avllm_clip_cls_rows(const void* ce, const void* pos, void* x, int32_t N, int32_t tokens, int32_t d, int32_t dtype, void* stream) { return av_clip_cls_rows(ce, pos, x, N, tokens, d, dtype, ST); }
int avllm_fuse_pool(const void* a, int32_t Ta, const void* v, int32_t Tv, const void* pe, int32_t P, void* out, int32_t B, int32_t L,
                    int32_t S_out, int32_t D, float fs, int32_t dtype, void* stream) { return av_fuse_pool(a, Ta, v, Tv, pe, P, out, B, L, S_out, D, fs, dtype, ST); }
int avllm_grad_sumsq(const float* g, int64_t n, float* sumsq, void* stream) { return av_grad_sumsq(g, n, sumsq, ST); }
int avllm_grad_sumsq_det(const float* g, int64_t n, float* partials, int32_t nparts, float* sumsq, void* stream) { return av_grad_sumsq_det(g, n, partials, nparts, sumsq, ST); }
int avllm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                     int32_t step, const float* sumsq, float max_norm, float prescale, const float* guard, float* skipped,
                     const avllm_step_state* state, void* stream) {
    return av_adamw_step(p, g, m, v, n, lr, b1, b2, eps, wd, step, sumsq, max_norm, prescale, guard, skipped, state, ST);
}
int avllm_act_residual(const void* x, const void* r, void* y, int64_t n, int32_t act, int32_t dtype, void* stream) { return av_act_residual(x, r, y, n, act, dtype, ST); }
int avllm_step_advance(avllm_step_state* state, const avllm_schedule* sched, void* stream) { return av_step_advance(state, sched, ST); }
int avllm_lora_pack(const float* A, const float* Bm, int32_t r, int32_t din, int32_t dout, void* A_pad, void* AT_pad, int64_t ld_at,
                    void* B_pad, void* BT_pad, int32_t dtype, void* stream) { return av_lora_pack(A, Bm, r, din, dout, A_pad, AT_pad, ld_at, B_pad, BT_pad, dtype, ST); }
}
