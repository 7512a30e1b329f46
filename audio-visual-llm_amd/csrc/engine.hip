// Model-level entry points: the kernel sequences for the Whisper encoder, the CLIP vision tower, and the
// Llama + LoRA forward / backward / KV-cache decode.  Host-side C++ only enqueues kernels on the caller's
// stream (no allocation, no sync), so a caller can capture any of these into a hipGraph.
//
// Reference call stacks: encode_audio clip_whisper_model.py:1067-1106 -> WhisperEncoder.forward
// HF:models/whisper/modeling_whisper.py:592-646; encode_video :1108-1146 -> CLIPVisionModel.forward
// HF:models/clip/modeling_clip.py:641-656; self.llm(...) :602-613 -> LlamaForCausalLM.forward
// HF:models/llama/modeling_llama.py:435-488; loss.backward() trainer/clip_whisper_trainer.py:454.
#include "common.h"
#include "avllm_internal.h"

namespace {

struct Bump {
    char* base; size_t cap; size_t off = 0; bool ok = true;
    Bump(void* p, size_t c) : base((char*)p), cap(c) {}
    void* take(size_t bytes) {
        const size_t a = (off + 255) & ~(size_t)255;
        if (a + bytes > cap) { ok = false; off = a + bytes; return base; }
        off = a + bytes;
        return base + a;
    }
};
// dry-run sizing uses a null base and an unlimited cap
inline size_t bump_size(const Bump& b) { return ((b.off + 255) & ~(size_t)255) + 256; }

inline avllm_gemm_desc gemm_desc(int dtype, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K) {
    avllm_gemm_desc g = {};
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.dtype = dtype; g.alpha = 1.0f;
    return g;
}

inline AvRopeScale llama_rope_scale(const avllm_llama* m) {
    AvRopeScale sc;
    sc.factor = m->rope_factor; sc.low_freq_factor = m->rope_low_freq_factor; sc.high_freq_factor = m->rope_high_freq_factor; sc.orig_ctx = m->rope_orig_ctx;
    return sc;
}

// ------------------------------------------------------------------ block-scaled fp8 projections (BASELINE config 5)
// Scratch for the quantised activation of one projection input: codes [M, Kmax] + its scale image.  One quantisation serves every
// projection that reads the same input (q, k and v of a decoder layer).
struct F8Buf { void* q = nullptr; void* s = nullptr; };
void carve_f8(Bump& b, F8Buf& f, long M, int kmax) {
    f.q = b.take((size_t)M * kmax);
    f.s = b.take(avllm_mx_scale_bytes((int)M, kmax));
}
inline int f8_quant(const F8Buf& f, const void* x, long ldx, int M, int K, hipStream_t st) {
    return av_mx_quantize(x, ldx, M, K, f.q, K, f.s, 0, AV_BF16, st);
}
// C = act(Aq.W8^T + bias) + R with the activation already quantised into f
inline int f8_proj(const F8Buf& f, int M, int K, const void* W8, const void* S8, int N, void* C, long ldc, const void* bias, int act, const void* R,
                   long ldr, hipStream_t st) {
    avllm_gemm_f8_desc d = {};
    d.A = f.q; d.SA = f.s; d.B = W8; d.SB = S8; d.C = C; d.bias = bias; d.R = R;
    d.lda = K; d.ldb = K; d.ldc = ldc; d.ldr = ldr; d.M = M; d.N = N; d.K = K; d.act = act;
    return av_gemm_f8(&d, st);
}

// ------------------------------------------------------------------ encoders
struct EncBuf { void *x, *xn, *qkv, *att, *ff; F8Buf f8; };

int encoder_layers(int dtype, const avllm_enc_layer* L, int layers, int d, int heads, int ffn, int tokens, long items,
                   float eps, int act, const EncBuf& b, bool cls_only_last, void* cls_out, void* xc, void* xcn, hipStream_t st, bool fp8 = false) {
    const long M = items * tokens;
    const int hd = d / heads;
    const size_t es = av_dtype_size(dtype);
    for (int l = 0; l < layers; ++l) {
        const avllm_enc_layer& P = L[l];
        AV_CHECK_ARG(!fp8 || (P.wqkv8 && P.sqkv8 && P.wo8 && P.so8 && P.w18 && P.s18 && P.w28 && P.s28), "encoder layer %d: fp8 mode without fp8 weight images", l);
        const bool nq = fp8 && d % 128 == 0 && d <= 8192 && !av_knob(AV_KNOB_F8_UNFUSED_QUANT);      // LayerNorm straight to e4m3 + scales (fp8.hip norm_mxq_kernel)
        if (nq) AV_TRY(av_norm_mxq(b.x, P.ln1_w, P.ln1_b, nullptr, nullptr, b.f8.q, d, b.f8.s, M, d, eps, st));
        else AV_TRY(av_layernorm(b.x, P.ln1_w, P.ln1_b, b.xn, M, d, eps, dtype, st));
        avllm_gemm_desc g;
        if (fp8) {
            if (!nq) AV_TRY(f8_quant(b.f8, b.xn, d, (int)M, d, st));
            AV_TRY(f8_proj(b.f8, (int)M, d, P.wqkv8, P.sqkv8, 3 * d, b.qkv, 3 * d, P.bqkv, AV_ACT_NONE, nullptr, 0, st));
        } else {
            g = gemm_desc(dtype, b.xn, d, P.wqkv, d, b.qkv, 3 * d, (int)M, 3 * d, d);
            g.bias = P.bqkv;
            AV_TRY(av_gemm(&g, st));
        }
        const char* qkv = (const char*)b.qkv;
        // fp8: the attention output is only ever the out-projection's A operand -- the one-pass kernel (CLIP: <= 272 tokens) block-scales it in its
        // epilogue; the bf16 tensor is not written (the CLS-only last block keeps the bf16 form)
        const bool att_q = fp8 && !(cls_only_last && l == layers - 1) && av_attention_fwd_mxq_ok((int)items, tokens, heads, hd, dtype, heads);
        if (att_q) AV_TRY(av_attention_fwd_mxq(qkv, qkv + (size_t)d * es, qkv + (size_t)2 * d * es, b.f8.q, d, b.f8.s, (int)items, tokens, heads, hd,
                                               3 * d, 3 * d, 3 * d, 1.0f / sqrtf((float)hd), st));
        else
        AV_TRY(av_attention_fwd(qkv, qkv + (size_t)d * es, qkv + (size_t)2 * d * es, b.att, nullptr, (int)items, tokens, tokens,
                                heads, hd, 3 * d, 3 * d, 3 * d, d, 1.0f / sqrtf((float)hd), 0, dtype, 0, st));
        if (cls_only_last && l == layers - 1) {
            // only token 0 of the last block is consumed (clip_whisper_model.py:1141): finish the block on CLS rows
            g = gemm_desc(dtype, b.att, (long)tokens * d, P.wo, d, xc, d, (int)items, d, d);
            g.bias = P.bo; g.R = b.x; g.ldr = (long)tokens * d;
            AV_TRY(av_gemm(&g, st));
            AV_TRY(av_layernorm(xc, P.ln2_w, P.ln2_b, xcn, items, d, eps, dtype, st));
            g = gemm_desc(dtype, xcn, d, P.w1, d, b.ff, ffn, (int)items, ffn, d);
            g.bias = P.b1; g.act = act;
            AV_TRY(av_gemm(&g, st));
            g = gemm_desc(dtype, b.ff, ffn, P.w2, ffn, cls_out, d, (int)items, d, ffn);
            g.bias = P.b2; g.R = xc; g.ldr = d;
            AV_TRY(av_gemm(&g, st));
            return AV_OK;
        }
        if (fp8) {      // the CLS-only last block above stays bf16: a handful of rows
            if (!att_q) AV_TRY(f8_quant(b.f8, b.att, d, (int)M, d, st));
            AV_TRY(f8_proj(b.f8, (int)M, d, P.wo8, P.so8, d, b.x, d, P.bo, AV_ACT_NONE, b.x, d, st));
            if (nq) AV_TRY(av_norm_mxq(b.x, P.ln2_w, P.ln2_b, nullptr, nullptr, b.f8.q, d, b.f8.s, M, d, eps, st));
            else {
                AV_TRY(av_layernorm(b.x, P.ln2_w, P.ln2_b, b.xn, M, d, eps, dtype, st));
                AV_TRY(f8_quant(b.f8, b.xn, d, (int)M, d, st));
            }
            {   // fc1 with its output quantised in the epilogue (codes + scale image live in the bf16 ff buffer's memory: 1 + 1/32 of its 2 bytes
                // per element), so fc2 reads them directly: no bf16 copy of the [M, ffn] activation, no quantiser pass over it
                avllm_gemm_f8_desc q1 = {};
                q1.A = b.f8.q; q1.SA = b.f8.s; q1.B = P.w18; q1.SB = P.s18; q1.bias = P.b1; q1.lda = d; q1.ldb = d; q1.M = (int)M; q1.N = ffn; q1.K = d; q1.act = act;
                F8Buf ffq;
                ffq.q = b.ff; ffq.s = (char*)b.ff + (((size_t)M * ffn + 255) & ~(size_t)255);
                q1.Cq = ffq.q; q1.SCq = ffq.s; q1.ldcq = ffn;
                if (!av_knob(AV_KNOB_F8_UNFUSED_QUANT) && avllm_gemm_f8_takes_quantised_output(&q1) && (size_t)M * ffn + 256 + avllm_mx_scale_bytes((int)M, ffn) <= (size_t)M * ffn * es) {
                    AV_TRY(av_gemm_f8(&q1, st));
                    AV_TRY(f8_proj(ffq, (int)M, ffn, P.w28, P.s28, d, b.x, d, P.b2, AV_ACT_NONE, b.x, d, st));
                    continue;
                }
            }
            AV_TRY(f8_proj(b.f8, (int)M, d, P.w18, P.s18, ffn, b.ff, ffn, P.b1, act, nullptr, 0, st));
            AV_TRY(f8_quant(b.f8, b.ff, ffn, (int)M, ffn, st));
            AV_TRY(f8_proj(b.f8, (int)M, ffn, P.w28, P.s28, d, b.x, d, P.b2, AV_ACT_NONE, b.x, d, st));
            continue;
        }
        g = gemm_desc(dtype, b.att, d, P.wo, d, b.x, d, (int)M, d, d);
        g.bias = P.bo; g.R = b.x; g.ldr = d;
        AV_TRY(av_gemm(&g, st));
        AV_TRY(av_layernorm(b.x, P.ln2_w, P.ln2_b, b.xn, M, d, eps, dtype, st));
        g = gemm_desc(dtype, b.xn, d, P.w1, d, b.ff, ffn, (int)M, ffn, d);
        g.bias = P.b1; g.act = act;
        AV_TRY(av_gemm(&g, st));
        g = gemm_desc(dtype, b.ff, ffn, P.w2, ffn, b.x, d, (int)M, d, ffn);
        g.bias = P.b2; g.R = b.x; g.ldr = d;
        AV_TRY(av_gemm(&g, st));
    }
    return AV_OK;
}

struct WhisperWs { void *cols1, *h1, *cols2; EncBuf e; };
void carve_whisper(const avllm_whisper* w, int B, Bump& b, WhisperWs& s) {
    const size_t es = av_dtype_size(w->dtype);
    const long T2 = 2L * w->n_ctx, M = (long)B * w->n_ctx;
    s.cols1 = b.take((size_t)B * T2 * w->k1pad * es);
    s.h1 = b.take((size_t)B * T2 * w->d * es);
    s.cols2 = b.take((size_t)M * 3 * w->d * es);
    s.e.x = b.take((size_t)M * w->d * es);
    s.e.xn = b.take((size_t)M * w->d * es);
    s.e.qkv = b.take((size_t)M * 3 * w->d * es);
    s.e.att = b.take((size_t)M * w->d * es);
    s.e.ff = b.take((size_t)M * w->ffn * es);
    if (w->fp8) carve_f8(b, s.e.f8, M, w->ffn > w->d ? w->ffn : w->d);
}

struct ClipWs { void* cols; EncBuf e; void *xc, *xcn; };
void carve_clip(const avllm_clip* c, int N, Bump& b, ClipWs& s, int& kpad) {
    const size_t es = av_dtype_size(c->dtype);
    const int g = c->image / c->patch;
    kpad = (3 * c->patch * c->patch + 63) / 64 * 64;
    const long M = (long)N * c->tokens;
    s.cols = b.take((size_t)N * g * g * kpad * es);
    s.e.x = b.take((size_t)M * c->d * es);
    s.e.xn = b.take((size_t)M * c->d * es);
    s.e.qkv = b.take((size_t)M * 3 * c->d * es);
    s.e.att = b.take((size_t)M * c->d * es);
    s.e.ff = b.take((size_t)M * c->ffn * es);
    s.xc = b.take((size_t)N * c->d * es);
    s.xcn = b.take((size_t)N * c->d * es);
    if (c->fp8) carve_f8(b, s.e.f8, M, c->ffn > c->d ? c->ffn : c->d);
}

// ------------------------------------------------------------------ llama
inline int llama_kv_heads(const avllm_llama* m);
inline int llama_dkv(const avllm_llama* m);
inline int llama_qw(const avllm_llama* m);
inline int llama_off(const avllm_llama* m, int j);
inline int llama_wid(const avllm_llama* m, int j);
struct LlamaLayerAct {
    void *xn1, *qkv, *att, *tqkv, *to, *h1, *gu;
    float *rstd1, *rstd2, *lse;
};
struct LlamaTrainWs {
    void** resid;            // host array [layers+1] (lives in a std::vector owned by the caller frame)
    LlamaLayerAct* act;      // host array [layers]
    void *xn2, *hmid, *xf, *logits, *xd;
    float *rstd_f, *row_lse, *delta, *rope_tab;
    // backward scratch
    void *dres, *dxn, *dgu, *dhmid, *dqkv, *datt, *dtqkv, *dto;
    F8Buf f8;                // fp8 mode: the quantised input of the projection being computed
};

void carve_llama_train(const avllm_llama* m, int B, int S, Bump& b, LlamaTrainWs& w, void** resid, LlamaLayerAct* act) {
    const size_t es = av_dtype_size(m->dtype);
    const long M = (long)B * S;
    const int d = m->d, f = m->ffn;
    w.resid = resid; w.act = act;
    for (int l = 0; l <= m->layers; ++l) resid[l] = b.take((size_t)M * d * es);
    for (int l = 0; l < m->layers; ++l) {
        LlamaLayerAct& a = act[l];
        a.xn1 = b.take((size_t)M * d * es);
        a.qkv = b.take((size_t)M * llama_qw(m) * es);
        a.att = b.take((size_t)M * d * es);
        a.tqkv = b.take((size_t)M * 3 * AVLLM_LORA_PAD * es);
        a.to = b.take((size_t)M * AVLLM_LORA_PAD * es);
        a.h1 = b.take((size_t)M * d * es);
        a.gu = b.take((size_t)M * 2 * f * es);
        a.rstd1 = (float*)b.take((size_t)M * 4);
        a.rstd2 = (float*)b.take((size_t)M * 4);
        a.lse = (float*)b.take((size_t)B * m->heads * S * 4);
    }
    w.xn2 = b.take((size_t)M * d * es);
    w.hmid = b.take((size_t)M * f * es);
    w.xf = b.take((size_t)M * d * es);
    w.logits = b.take((size_t)M * m->vocab * es);
    w.xd = b.take((size_t)M * d * es);
    w.rstd_f = (float*)b.take((size_t)M * 4);
    w.row_lse = (float*)b.take((size_t)M * 4);
    w.delta = (float*)b.take((size_t)B * m->heads * S * 4);
    w.rope_tab = (float*)b.take((size_t)S * (d / m->heads) * 4);
    w.dres = b.take((size_t)M * d * es);
    w.dxn = b.take((size_t)M * d * es);
    w.dgu = b.take((size_t)M * 2 * f * es);
    w.dhmid = b.take((size_t)M * f * es);
    w.dqkv = b.take((size_t)M * llama_qw(m) * es);
    w.datt = b.take((size_t)M * d * es);
    w.dtqkv = b.take((size_t)M * 3 * AVLLM_LORA_PAD * es);
    w.dto = b.take((size_t)M * AVLLM_LORA_PAD * es);
    if (m->fp8) carve_f8(b, w.f8, M, f > d ? f : d);
}

int check_llama(const avllm_llama* m) {
    AV_CHECK_ARG(m && m->layer && m->embed && m->norm_w && m->lm_head, "llama: null model fields");
    AV_CHECK_ARG(m->d % m->heads == 0 && m->d % 64 == 0 && m->ffn % 64 == 0, "llama: d=%d ffn=%d must be multiples of 64", m->d, m->ffn);
    AV_CHECK_ARG(m->layers > 0 && m->layers <= 256, "llama: layers=%d", m->layers);
    AV_CHECK_ARG(m->kv_heads >= 0 && (m->kv_heads == 0 || m->heads % m->kv_heads == 0), "llama: heads=%d kv_heads=%d", m->heads, m->kv_heads);
    return AV_OK;
}
// grouped-query geometry: q is d wide, k and v are dkv = kv_heads*hd wide; the fused row is [q | k | v] = qw columns
inline int llama_kv_heads(const avllm_llama* m) { return m->kv_heads > 0 ? m->kv_heads : m->heads; }
inline int llama_dkv(const avllm_llama* m) { return llama_kv_heads(m) * (m->d / m->heads); }
inline int llama_qw(const avllm_llama* m) { return m->d + 2 * llama_dkv(m); }
inline int llama_off(const avllm_llama* m, int j) { return j == 0 ? 0 : (j == 1 ? m->d : m->d + llama_dkv(m)); }     // column (= wqkv row) of slice j
inline int llama_wid(const avllm_llama* m, int j) { return j == 0 ? m->d : llama_dkv(m); }

// y[:, slice j] = x W_j^T (+ t_j B_j^T)
int lora_proj(const avllm_llama* m, const void* x, long ldx, const void* W, long ldw, int K, int N, const avllm_lora_mod& lm,
              void* t, long ldt, void* y, long ldy, const void* R, long ldr, int M, hipStream_t st, const void* xl = nullptr,
              uint32_t a_seed = 0, float a_p = 0.f) {
    // (the step's base seed may live in device memory: m->dropout_seed_dev, see avllm_step_state)
    avllm_gemm_desc g;
    const bool has = lm.A_pad != nullptr;
    if (has) {
        // xl = dropout(x) for the adapter branch when lora_dropout is active (peft: lora_B(lora_A(dropout(x))))
        g = gemm_desc(m->dtype, xl ? xl : x, xl ? (long)K : ldx, lm.A_pad, K, t, ldt, M, AVLLM_LORA_PAD, K);
        g.alpha = m->lora_scale;
        g.a_drop_seed = a_seed; g.a_drop_p = a_p; g.seed_dev = m->dropout_seed_dev;      // bf16: dropout generated inside the rank-side GEMM
        g.n_valid = m->lora_r;                         // rank padded to 64: the padding columns are written as zeros, not computed
        AV_TRY(av_gemm(&g, st));
    }
    g = gemm_desc(m->dtype, x, ldx, W, ldw, y, ldy, M, N, K);
    if (has) { g.A2 = t; g.lda2 = ldt; g.B2 = lm.B_pad; g.ldb2 = AVLLM_LORA_PAD; g.K2 = AVLLM_LORA_PAD; }
    g.R = R; g.ldr = ldr;
    return av_gemm(&g, st);
}

// fp8 mode: the adapter term alone, y += scale * (dropout(x) A^T) B^T on top of a base product already in y (bf16, rank-side GEMM + K=64 GEMM)
int lora_add(const avllm_llama* m, const void* x, long ldx, int K, int N, const avllm_lora_mod& lm, void* t, long ldt, void* y, long ldy, int M,
             hipStream_t st, const void* xl, uint32_t a_seed, float a_p) {
    if (!lm.A_pad) return AV_OK;
    avllm_gemm_desc g = gemm_desc(m->dtype, xl ? xl : x, xl ? (long)K : ldx, lm.A_pad, K, t, ldt, M, AVLLM_LORA_PAD, K);
    g.alpha = m->lora_scale; g.a_drop_seed = a_seed; g.a_drop_p = a_p; g.seed_dev = m->dropout_seed_dev; g.n_valid = m->lora_r;
    AV_TRY(av_gemm(&g, st));
    g = gemm_desc(m->dtype, t, ldt, lm.B_pad, AVLLM_LORA_PAD, y, ldy, M, N, AVLLM_LORA_PAD);
    g.R = y; g.ldr = ldy;
    return av_gemm(&g, st);
}

}  // namespace

// =============================================================================================== Whisper
extern "C" size_t avllm_whisper_workspace_bytes(const avllm_whisper* w, int32_t B) {
    Bump b(nullptr, (size_t)-1);
    WhisperWs s;
    carve_whisper(w, B, b, s);
    return bump_size(b);
}

extern "C" int avllm_whisper_encoder_fwd(const avllm_whisper* w, const float* mel, int32_t B, void* out, void* ws,
                                         size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AV_CHECK_ARG(w && mel && out && ws && B > 0, "whisper_encoder_fwd: null/empty");
    AV_CHECK_ARG(w->d % w->heads == 0 && w->d % 64 == 0 && w->ffn % 64 == 0 && w->k1pad % 64 == 0 && w->k1pad >= 3 * w->n_mels,
                 "whisper: d=%d ffn=%d k1pad=%d unsupported", w->d, w->ffn, w->k1pad);
    Bump b(ws, ws_bytes);
    WhisperWs s;
    carve_whisper(w, B, b, s);
    if (!b.ok) return av_set_error(AV_ERR_WORKSPACE, "whisper_encoder_fwd: workspace %zu < %zu bytes", ws_bytes, bump_size(b));
    const int dt = w->dtype, d = w->d, T2 = 2 * w->n_ctx;
    const long M = (long)B * w->n_ctx;
    AV_TRY(av_whisper_im2col1(mel, s.cols1, B, w->n_mels, T2, w->k1pad, dt, st));
    avllm_gemm_desc g = gemm_desc(dt, s.cols1, w->k1pad, w->conv1_w, w->k1pad, s.h1, d, B * T2, d, w->k1pad);
    g.bias = w->conv1_b; g.act = AV_ACT_GELU;
    AV_TRY(av_gemm(&g, st));
    AV_TRY(av_whisper_im2col2(s.h1, s.cols2, B, T2, d, dt, st));
    g = gemm_desc(dt, s.cols2, 3 * d, w->conv2_w, 3 * d, s.e.x, d, (int)M, d, 3 * d);
    g.bias = w->conv2_b; g.act = AV_ACT_GELU; g.R = w->pos; g.ldr = d; g.r_mod = w->n_ctx;
    AV_TRY(av_gemm(&g, st));
    AV_CHECK_ARG(!w->fp8 || (dt == AV_BF16 && d % 128 == 0 && w->ffn % 128 == 0), "whisper: fp8 needs bf16 activations and widths that are multiples of 128");
    AV_TRY(encoder_layers(dt, w->layer, w->layers, d, w->heads, w->ffn, w->n_ctx, B, 1e-5f, AV_ACT_GELU, s.e, false, nullptr,
                          nullptr, nullptr, st, w->fp8 != 0));
    return av_layernorm(s.e.x, w->lnf_w, w->lnf_b, out, M, d, 1e-5f, dt, st);
}

// =============================================================================================== CLIP
extern "C" size_t avllm_clip_workspace_bytes(const avllm_clip* c, int32_t N) {
    Bump b(nullptr, (size_t)-1);
    ClipWs s; int kpad;
    carve_clip(c, N, b, s, kpad);
    return bump_size(b);
}

extern "C" int avllm_clip_vision_cls_fwd(const avllm_clip* c, const void* frames, int32_t N, void* cls, void* ws,
                                         size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AV_CHECK_ARG(c && frames && cls && ws && N > 0, "clip_vision_cls_fwd: null/empty");
    const int g1 = c->image / c->patch;
    AV_CHECK_ARG(c->tokens == g1 * g1 + 1 && c->d % c->heads == 0 && c->d % 64 == 0 && c->ffn % 64 == 0,
                 "clip: tokens=%d d=%d ffn=%d inconsistent", c->tokens, c->d, c->ffn);
    Bump b(ws, ws_bytes);
    ClipWs s; int kpad;
    carve_clip(c, N, b, s, kpad);
    if (!b.ok) return av_set_error(AV_ERR_WORKSPACE, "clip_vision_cls_fwd: workspace %zu < %zu bytes", ws_bytes, bump_size(b));
    const int dt = c->dtype, d = c->d, np = g1 * g1;
    const size_t es = av_dtype_size(dt);
    AV_TRY(av_clip_patchify(frames, s.cols, N, c->image, c->patch, kpad, dt, st, c->frames_bf16 ? AV_BF16 : AV_F32));
    // patch embedding + position embedding of the patch tokens, scattered to rows 1.. of each frame
    avllm_gemm_desc g = gemm_desc(dt, s.cols, kpad, c->patch_w, kpad, s.e.xn, d, N * np, d, kpad);
    g.R = (const char*)c->pos + (size_t)d * es; g.ldr = d; g.r_mod = np;
    g.g_in = np; g.g_out = c->tokens; g.g_off = 1;
    AV_TRY(av_gemm(&g, st));
    AV_TRY(av_clip_cls_rows(c->class_emb, c->pos, s.e.xn, N, c->tokens, d, dt, st));
    AV_TRY(av_layernorm(s.e.xn, c->pre_ln_w, c->pre_ln_b, s.e.x, (long)N * c->tokens, d, c->eps, dt, st));
    AV_CHECK_ARG(!c->fp8 || (dt == AV_BF16 && d % 128 == 0 && c->ffn % 128 == 0), "clip: fp8 needs bf16 activations and widths that are multiples of 128");
    return encoder_layers(dt, c->layer, c->layers, d, c->heads, c->ffn, c->tokens, N, c->eps, AV_ACT_QUICK_GELU, s.e, true, cls,
                          s.xc, s.xcn, st, c->fp8 != 0);
}

// =============================================================================================== Llama train
extern "C" size_t avllm_llama_train_workspace_bytes(const avllm_llama* m, int32_t B, int32_t S) {
    Bump b(nullptr, (size_t)-1);
    LlamaTrainWs w;
    void* resid[257]; LlamaLayerAct act[256];
    if (!m || m->layers > 256) return 0;
    carve_llama_train(m, B, S, b, w, resid, act);
    return bump_size(b);
}

extern "C" int avllm_llama_lora_fwd_loss(const avllm_llama* m, const void* x, const int64_t* labels, int32_t B, int32_t S,
                                         void* logits_out, float* loss_sum, float* count, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AV_TRY(check_llama(m));
    AV_CHECK_ARG(x && ws && B > 0 && S > 0, "llama_lora_fwd_loss: null/empty");
    AV_CHECK_ARG(!labels || (loss_sum && count), "llama_lora_fwd_loss: labels need loss_sum/count");
    Bump b(ws, ws_bytes);
    LlamaTrainWs w;
    void* resid[257]; LlamaLayerAct act[256];
    carve_llama_train(m, B, S, b, w, resid, act);
    if (!b.ok) return av_set_error(AV_ERR_WORKSPACE, "llama_lora_fwd_loss: workspace %zu < %zu bytes", ws_bytes, bump_size(b));
    const int dt = m->dtype, d = m->d, f = m->ffn, H = m->heads, hd = d / H;
    const int Hkv = llama_kv_heads(m), dkv = llama_dkv(m), qw = llama_qw(m);
    const size_t es = av_dtype_size(dt);
    const int M = B * S;
    AV_HIP(hipMemcpyAsync(resid[0], x, (size_t)M * d * es, hipMemcpyDeviceToDevice, st));
    AV_TRY(av_rope_table(w.rope_tab, S, hd, 0, m->theta, st, nullptr, llama_rope_scale(m)));
    const bool drop = m->lora_dropout > 0.f;
    // bf16 (MFMA kernels): masks are generated inside the rank-side GEMMs; fp32 parity mode materialises dropout(x)
    const bool fuse_drop = drop && m->dtype == AV_BF16 && d % 256 == 0 && m->lora_r <= 16;
    const bool fp8 = m->fp8 != 0;
    AV_CHECK_ARG(!fp8 || (dt == AV_BF16 && d % 128 == 0 && f % 128 == 0 && m->vocab % 8 == 0 && m->lm_head8 && m->slm_head8),
                 "llama: fp8 needs bf16 activations, d and ffn multiples of 128 and the fp8 weight images");
    for (int l = 0; l < m->layers; ++l) {
        const avllm_llama_layer& P = m->layer[l];
        LlamaLayerAct& a = act[l];
        AV_TRY(av_rmsnorm_fwd(resid[l], P.ln1_w, a.xn1, a.rstd1, M, d, m->eps, dt, st));
        if (fp8) {      // one fp8 product for q|k|v (one quantisation of the normed input), adapters added on top
            AV_CHECK_ARG(P.wqkv8 && P.sqkv8 && P.wo8 && P.so8 && P.wgu8 && P.sgu8 && P.wdown8 && P.sdown8, "llama layer %d: fp8 mode without fp8 weight images", l);
            AV_TRY(f8_quant(w.f8, a.xn1, d, M, d, st));
            AV_TRY(f8_proj(w.f8, M, d, P.wqkv8, P.sqkv8, qw, a.qkv, qw, nullptr, AV_ACT_NONE, nullptr, 0, st));
        }
        // all three rank-side products t_j = s * dropout_j(xn1) A_j^T in one launch (xn1 read once): csrc/lora_batch.hip
        const bool batch_qkv = !fp8 && dt == AV_BF16 && P.lora[0].A_pad && P.lora[1].A_pad && P.lora[2].A_pad && (!drop || fuse_drop) &&
                               d % 256 == 0 && av_lora_batch_supported(dt, m->lora_r, 3) && !av_knob(AV_KNOB_LORA_UNBATCHED);
        if (batch_qkv) {
            const void* Ap[3] = {a.xn1, a.xn1, a.xn1}; const long la[3] = {d, d, d}; const int Kk[3] = {d, d, d};
            const void* Bp[3] = {P.lora[0].A_pad, P.lora[1].A_pad, P.lora[2].A_pad}; const long lb[3] = {d, d, d};
            void* Cp[3]; long lc[3]; uint32_t sd[3];
            for (int j = 0; j < 3; ++j) { Cp[j] = (char*)a.tqkv + (size_t)j * AVLLM_LORA_PAD * es; lc[j] = 3 * AVLLM_LORA_PAD; sd[j] = m->dropout_seed + 4u * l + j; }
            AV_TRY(av_lora_rank3(Ap, la, Kk, Bp, lb, Cp, lc, sd, 3, M, m->lora_r, m->lora_scale, drop ? m->lora_dropout : 0.f, m->dropout_seed_dev, 1, dt, st));
        }
        for (int j = 0; j < 3; ++j) {
            const void* xl = nullptr;
            const uint32_t sj = m->dropout_seed + 4u * l + j;
            if (drop && !fuse_drop && P.lora[j].A_pad) { AV_TRY(av_dropout(a.xn1, w.xd, M, d, sj, m->lora_dropout, dt, st, m->dropout_seed_dev)); xl = w.xd; }
            if (batch_qkv) {      // the projection with the adapter term as its second K segment; t_j is already there
                avllm_gemm_desc gp = gemm_desc(dt, a.xn1, d, (const char*)P.wqkv + (size_t)llama_off(m, j) * d * es, d,
                                               (char*)a.qkv + (size_t)llama_off(m, j) * es, qw, M, llama_wid(m, j), d);
                gp.A2 = (char*)a.tqkv + (size_t)j * AVLLM_LORA_PAD * es; gp.lda2 = 3 * AVLLM_LORA_PAD; gp.B2 = P.lora[j].B_pad; gp.ldb2 = AVLLM_LORA_PAD; gp.K2 = AVLLM_LORA_PAD;
                AV_TRY(av_gemm(&gp, st));
                continue;
            }
            if (fp8) {
                AV_TRY(lora_add(m, a.xn1, d, d, llama_wid(m, j), P.lora[j], (char*)a.tqkv + (size_t)j * AVLLM_LORA_PAD * es, 3 * AVLLM_LORA_PAD,
                                (char*)a.qkv + (size_t)llama_off(m, j) * es, qw, M, st, xl, sj, fuse_drop ? m->lora_dropout : 0.f));
                continue;
            }
            AV_TRY(lora_proj(m, a.xn1, d, (const char*)P.wqkv + (size_t)llama_off(m, j) * d * es, d, d, llama_wid(m, j), P.lora[j],
                             (char*)a.tqkv + (size_t)j * AVLLM_LORA_PAD * es, 3 * AVLLM_LORA_PAD,
                             (char*)a.qkv + (size_t)llama_off(m, j) * es, qw, nullptr, 0, M, st, xl, sj, fuse_drop ? m->lora_dropout : 0.f));
        }
        AV_TRY(av_rope_tab(a.qkv, qw, M, S, H + Hkv, hd, w.rope_tab, 0, dt, st));      // q and k slices are adjacent: H + Hkv heads
        const char* qkv = (const char*)a.qkv;
        AV_TRY(av_attention_fwd(qkv, qkv + (size_t)d * es, qkv + (size_t)(d + dkv) * es, a.att, a.lse, B, S, S, H, hd, qw, qw,
                                qw, d, 1.0f / sqrtf((float)hd), 1, dt, 0, st, Hkv));
        {
            const void* xl = nullptr;
            const uint32_t so = m->dropout_seed + 4u * l + 3;
            if (drop && !fuse_drop && P.lora[3].A_pad) { AV_TRY(av_dropout(a.att, w.xd, M, d, so, m->lora_dropout, dt, st, m->dropout_seed_dev)); xl = w.xd; }
            if (fp8) {
                AV_TRY(f8_quant(w.f8, a.att, d, M, d, st));
                AV_TRY(f8_proj(w.f8, M, d, P.wo8, P.so8, d, a.h1, d, nullptr, AV_ACT_NONE, resid[l], d, st));
                AV_TRY(lora_add(m, a.att, d, d, d, P.lora[3], a.to, AVLLM_LORA_PAD, a.h1, d, M, st, xl, so, fuse_drop ? m->lora_dropout : 0.f));
            } else
            AV_TRY(lora_proj(m, a.att, d, P.wo, d, d, d, P.lora[3], a.to, AVLLM_LORA_PAD, a.h1, d, resid[l], d, M, st, xl, so,
                             fuse_drop ? m->lora_dropout : 0.f));
        }
        AV_TRY(av_rmsnorm_fwd(a.h1, P.ln2_w, w.xn2, a.rstd2, M, d, m->eps, dt, st));
        if (fp8) {
            AV_TRY(f8_quant(w.f8, w.xn2, d, M, d, st));
            AV_TRY(f8_proj(w.f8, M, d, P.wgu8, P.sgu8, 2 * f, a.gu, 2 * f, nullptr, AV_ACT_NONE, nullptr, 0, st));
            AV_TRY(av_swiglu_fwd(a.gu, w.hmid, M, f, dt, st));
            AV_TRY(f8_quant(w.f8, w.hmid, f, M, f, st));
            AV_TRY(f8_proj(w.f8, M, f, P.wdown8, P.sdown8, d, resid[l + 1], d, nullptr, AV_ACT_NONE, a.h1, d, st));
            continue;
        }
        avllm_gemm_desc g = gemm_desc(dt, w.xn2, d, P.wgu, d, a.gu, 2 * f, M, 2 * f, d);
        AV_TRY(av_gemm(&g, st));
        AV_TRY(av_swiglu_fwd(a.gu, w.hmid, M, f, dt, st));
        g = gemm_desc(dt, w.hmid, f, P.wdown, f, resid[l + 1], d, M, d, f);
        g.R = a.h1; g.ldr = d;
        AV_TRY(av_gemm(&g, st));
    }
    AV_TRY(av_rmsnorm_fwd(resid[m->layers], m->norm_w, w.xf, w.rstd_f, M, d, m->eps, dt, st));
    avllm_gemm_desc g = gemm_desc(dt, w.xf, d, m->lm_head, d, w.logits, m->vocab, M, m->vocab, d);
    if (fp8) {
        AV_TRY(f8_quant(w.f8, w.xf, d, M, d, st));
        AV_TRY(f8_proj(w.f8, M, d, m->lm_head8, m->slm_head8, m->vocab, w.logits, m->vocab, nullptr, AV_ACT_NONE, nullptr, 0, st));
    } else
    AV_TRY(av_gemm(&g, st));
    if (logits_out) AV_HIP(hipMemcpyAsync(logits_out, w.logits, (size_t)M * m->vocab * es, hipMemcpyDeviceToDevice, st));
    if (labels) AV_TRY(av_ce_fwd(w.logits, m->vocab, labels, B, S, m->vocab, w.row_lse, loss_sum, count, dt, st));
    return AV_OK;
}

extern "C" int avllm_llama_lora_bwd(const avllm_llama* m, const int64_t* labels, int32_t B, int32_t S, const float* count,
                                    float grad_scale, void* ws, size_t ws_bytes, avllm_layer_cb after_layer, void* user,
                                    void* stream) {
    AV_TRY(check_llama(m));
    return avllm_llama_lora_bwd_layers(m, labels, B, S, count, grad_scale, ws, ws_bytes, m->layers - 1, 0, after_layer, user, stream);
}

extern "C" int avllm_llama_lora_bwd_layers(const avllm_llama* m, const int64_t* labels, int32_t B, int32_t S, const float* count,
                                           float grad_scale, void* ws, size_t ws_bytes, int32_t layer_hi, int32_t layer_lo,
                                           avllm_layer_cb after_layer, void* user, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AV_TRY(check_llama(m));
    AV_CHECK_ARG(layer_lo >= 0 && layer_lo <= layer_hi && layer_hi < m->layers, "llama_lora_bwd_layers: bad layer range [%d, %d]", layer_lo, layer_hi);
    AV_CHECK_ARG(labels && count && ws && m->lm_head_t, "llama_lora_bwd: null (training needs the transposed weight images)");
    Bump b(ws, ws_bytes);
    LlamaTrainWs w;
    void* resid[257]; LlamaLayerAct act[256];
    carve_llama_train(m, B, S, b, w, resid, act);
    if (!b.ok) return av_set_error(AV_ERR_WORKSPACE, "llama_lora_bwd: workspace %zu < %zu bytes", ws_bytes, bump_size(b));
    const int dt = m->dtype, d = m->d, f = m->ffn, H = m->heads, hd = d / H, V = m->vocab;
    const int Hkv = llama_kv_heads(m), dkv = llama_dkv(m), qw = llama_qw(m);
    const size_t es = av_dtype_size(dt);
    const int M = B * S, R = m->lora_r;
    const float sc = m->lora_scale;
    const bool drop = m->lora_dropout > 0.f;
    // bf16 (MFMA kernels): masks are generated inside the rank-side GEMMs; fp32 parity mode materialises dropout(x)
    const bool fuse_drop = drop && m->dtype == AV_BF16 && d % 256 == 0 && m->lora_r <= 16;
    avllm_gemm_desc g;
    if (layer_hi == m->layers - 1) {      // the piece that starts at the top also runs loss -> lm_head -> final norm
        AV_TRY(av_ce_bwd(w.logits, V, labels, w.row_lse, count, grad_scale, w.logits, B, S, V, dt, st));
        g = gemm_desc(dt, w.logits, V, m->lm_head_t, V, w.dxn, d, M, d, V);
        AV_CHECK_ARG(V % 64 == 0, "llama_lora_bwd: vocab %d must be a multiple of 64", V);
        AV_TRY(av_gemm(&g, st));
        AV_TRY(av_rmsnorm_bwd(w.dxn, resid[m->layers], m->norm_w, w.rstd_f, nullptr, w.dres, M, d, dt, st));
    }
    for (int l = layer_hi; l >= layer_lo; --l) {
        const avllm_llama_layer& P = m->layer[l];
        LlamaLayerAct& a = act[l];
        AV_CHECK_ARG(P.wqkv_t && P.wo_t && P.wgu_t && P.wdown_t, "llama_lora_bwd: layer %d has no transposed weights", l);
        // ---- MLP: resid[l+1] = h1 + down(silu(g)*u)
        g = gemm_desc(dt, w.dres, d, P.wdown_t, d, w.dhmid, f, M, f, d);
        AV_TRY(av_gemm(&g, st));
        AV_TRY(av_swiglu_bwd(w.dhmid, a.gu, w.dgu, M, f, dt, st));
        g = gemm_desc(dt, w.dgu, 2 * f, P.wgu_t, 2 * f, w.dxn, d, M, d, 2 * f);
        AV_TRY(av_gemm(&g, st));
        AV_TRY(av_rmsnorm_bwd(w.dxn, a.h1, P.ln2_w, a.rstd2, w.dres, w.dres, M, d, dt, st));      // dres = d h1
        // ---- o_proj (+LoRA): h1 = resid[l] + att Wo^T + to Bo^T
        const avllm_lora_mod& lo = P.lora[3];
        g = gemm_desc(dt, w.dres, d, P.wo_t, d, w.datt, d, M, d, d);
        if (lo.A_pad) {
            AV_TRY(av_gemm_tn(w.dres, d, d, a.to, AVLLM_LORA_PAD, R, M, lo.gB, R, 1.0f, dt, st));
            avllm_gemm_desc gt = gemm_desc(dt, w.dres, d, lo.BT_pad, d, w.dto, AVLLM_LORA_PAD, M, AVLLM_LORA_PAD, d);
            gt.alpha = sc; gt.n_valid = R;
            AV_TRY(av_gemm(&gt, st));
            const void* xin = a.att;
            if (drop && !fuse_drop) { AV_TRY(av_dropout(a.att, w.xd, M, d, m->dropout_seed + 4u * l + 3, m->lora_dropout, dt, st, m->dropout_seed_dev)); xin = w.xd; }
            AV_TRY(av_gemm_tn(w.dto, AVLLM_LORA_PAD, R, xin, d, d, M, lo.gA, d, 1.0f, dt, st, m->dropout_seed + 4u * l + 3,
                              fuse_drop ? m->lora_dropout : 0.f, m->dropout_seed_dev));
            if (!drop) { g.A2 = w.dto; g.lda2 = AVLLM_LORA_PAD; g.B2 = lo.AT_pad; g.ldb2 = lo.ld_at; g.K2 = AVLLM_LORA_PAD; }
        }
        AV_TRY(av_gemm(&g, st));
        if (lo.A_pad && drop) {       // d att += mask_o * (dto . A_o) / (1-p): the adapter's input gradient passes back through its dropout
            const void* Tp[1] = {w.dto}; const void* Ap[1] = {lo.AT_pad};
            const long lt[1] = {AVLLM_LORA_PAD}, la[1] = {lo.ld_at};
            const uint32_t sd[1] = {m->dropout_seed + 4u * l + 3};
            if (fuse_drop && av_lora_dx_masked_supported(dt, d, R, lt, la, 1, d, d)) {
                AV_TRY(av_lora_dx_masked(Tp, lt, Ap, la, sd, 1, R, w.datt, d, w.datt, d, M, d, m->lora_dropout, m->dropout_seed_dev, dt, st));
            } else {
            avllm_gemm_desc gm = gemm_desc(dt, w.dto, AVLLM_LORA_PAD, lo.AT_pad, lo.ld_at, w.datt, d, M, d, AVLLM_LORA_PAD);
            gm.R = w.datt; gm.ldr = d; gm.drop_seed = m->dropout_seed + 4u * l + 3; gm.drop_p = m->lora_dropout; gm.seed_dev = m->dropout_seed_dev;
            AV_TRY(av_gemm(&gm, st));
            }
        }
        // ---- attention
        const char* qkv = (const char*)a.qkv;
        char* dqkv = (char*)w.dqkv;
        const bool fuse_rope = av_attention_bwd_fuses_rope(dt, hd, 0);      // bf16: the inverse RoPE rides in the dq/dk epilogues
        AV_TRY(av_attention_bwd(qkv, qkv + (size_t)d * es, qkv + (size_t)(d + dkv) * es, a.att, w.datt, a.lse, dqkv, dqkv + (size_t)d * es,
                                dqkv + (size_t)(d + dkv) * es, w.delta, B, S, H, hd, qw, qw, qw, d, qw, qw, qw,
                                1.0f / sqrtf((float)hd), 1, dt, 0, st, Hkv, fuse_rope ? w.rope_tab : nullptr));
        if (!fuse_rope) AV_TRY(av_rope_tab(dqkv, qw, M, S, H + Hkv, hd, w.rope_tab, 1, dt, st));
        // ---- q,k,v projections (+LoRA)
        bool any = false, contiguous = true;
        const bool batch_bwd = dt == AV_BF16 && P.lora[0].A_pad && P.lora[1].A_pad && P.lora[2].A_pad && (!drop || fuse_drop) && d % 256 == 0 &&
                               dkv % 256 == 0 && av_lora_batch_supported(dt, R, 3) && !av_knob(AV_KNOB_LORA_UNBATCHED);
        if (batch_bwd) {      // three launches for the three adapters' dB, dt and dA (csrc/lora_batch.hip) instead of nine
            any = true;
            const void* Tq[3]; long ltq[3]; float* gBp[3]; long lgb[3]; int c0[3], nc[3];
            const void* dyp[3]; long ldy[3]; int Kd[3]; const void* BTp[3]; long lbt[3]; void* dtp[3]; long ldt3[3];
            const void* dtc[3]; float* gAp[3]; long lga[3]; uint32_t sd[3];
            for (int j = 0; j < 3; ++j) {
                const avllm_lora_mod& lj = P.lora[j];
                Tq[j] = (char*)a.tqkv + (size_t)j * AVLLM_LORA_PAD * es; ltq[j] = 3 * AVLLM_LORA_PAD; gBp[j] = lj.gB; lgb[j] = R;
                c0[j] = llama_off(m, j); nc[j] = llama_wid(m, j);
                dyp[j] = dqkv + (size_t)llama_off(m, j) * es; ldy[j] = qw; Kd[j] = llama_wid(m, j); BTp[j] = lj.BT_pad; lbt[j] = llama_wid(m, j);
                dtp[j] = (char*)w.dtqkv + (size_t)j * AVLLM_LORA_PAD * es; ldt3[j] = 3 * AVLLM_LORA_PAD; dtc[j] = dtp[j];
                gAp[j] = lj.gA; lga[j] = d; sd[j] = m->dropout_seed + 4u * l + j;
                if (lj.ld_at != 3 * AVLLM_LORA_PAD ||
                    (const char*)lj.AT_pad != (const char*)P.lora[0].AT_pad + (size_t)j * AVLLM_LORA_PAD * es) contiguous = false;
            }
            AV_TRY(av_gemm_tn_multi(dqkv, qw, qw, Tq, ltq, gBp, lgb, c0, nc, nullptr, 3, R, M, 1.0f, 0.f, nullptr, 0, dt, st));          // dB_j = dy_j^T t_j
            AV_TRY(av_lora_rank3(dyp, ldy, Kd, BTp, lbt, dtp, ldt3, nullptr, 3, M, R, sc, 0.f, nullptr, 0, dt, st));                     // dt_j = s dy_j B_j
            AV_TRY(av_gemm_tn_multi(a.xn1, d, d, dtc, ldt3, gAp, lga, nullptr, nullptr, sd, 3, R, M, 1.0f, drop ? m->lora_dropout : 0.f,   // dA_j = dt_j^T dropout_j(xn1)
                                    m->dropout_seed_dev, 1, dt, st));
        } else
        for (int j = 0; j < 3; ++j) {
            const avllm_lora_mod& lj = P.lora[j];
            if (!lj.A_pad) { contiguous = false; continue; }
            any = true;
            const char* dy = dqkv + (size_t)llama_off(m, j) * es;
            const int wj = llama_wid(m, j);
            char* dtj = (char*)w.dtqkv + (size_t)j * AVLLM_LORA_PAD * es;
            AV_TRY(av_gemm_tn(dy, qw, wj, (char*)a.tqkv + (size_t)j * AVLLM_LORA_PAD * es, 3 * AVLLM_LORA_PAD, R, M, lj.gB, R, 1.0f, dt, st));
            avllm_gemm_desc gt = gemm_desc(dt, dy, qw, lj.BT_pad, wj, dtj, 3 * AVLLM_LORA_PAD, M, AVLLM_LORA_PAD, wj);
            gt.alpha = sc; gt.n_valid = R;
            AV_TRY(av_gemm(&gt, st));
            const void* xin = a.xn1;
            if (drop && !fuse_drop) { AV_TRY(av_dropout(a.xn1, w.xd, M, d, m->dropout_seed + 4u * l + j, m->lora_dropout, dt, st, m->dropout_seed_dev)); xin = w.xd; }
            AV_TRY(av_gemm_tn(dtj, 3 * AVLLM_LORA_PAD, R, xin, d, d, M, lj.gA, d, 1.0f, dt, st, m->dropout_seed + 4u * l + j,
                              fuse_drop ? m->lora_dropout : 0.f, m->dropout_seed_dev));
            if (lj.ld_at != 3 * AVLLM_LORA_PAD ||
                (const char*)lj.AT_pad != (const char*)P.lora[0].AT_pad + (size_t)j * AVLLM_LORA_PAD * es) contiguous = false;
        }
        if (l > 0) {      // d(inputs_embeds) is not needed: encoders/connectors are frozen
            AV_CHECK_ARG(!any || contiguous || drop, "llama_lora_bwd: q/k/v AT_pad images must be the three 64-column slices of one [d,192] matrix");
            g = gemm_desc(dt, w.dqkv, qw, P.wqkv_t, qw, w.dxn, d, M, d, qw);
            if (any && !drop) { g.A2 = w.dtqkv; g.lda2 = 3 * AVLLM_LORA_PAD; g.B2 = P.lora[0].AT_pad; g.ldb2 = 3 * AVLLM_LORA_PAD; g.K2 = 3 * AVLLM_LORA_PAD; }
            AV_TRY(av_gemm(&g, st));
            if (any && drop) {
                // all three adapters' input gradients in one pass over dX when the fused kernel applies (csrc/lora_dx.hip)
                const void* Tp[3]; const void* Ap[3]; long lt[3], la[3]; uint32_t sd[3]; int nj = 0;
                for (int j = 0; j < 3; ++j) {
                    if (!P.lora[j].A_pad) continue;
                    Tp[nj] = (char*)w.dtqkv + (size_t)j * AVLLM_LORA_PAD * es; lt[nj] = 3 * AVLLM_LORA_PAD;
                    Ap[nj] = P.lora[j].AT_pad; la[nj] = P.lora[j].ld_at; sd[nj] = m->dropout_seed + 4u * l + j; ++nj;
                }
                if (fuse_drop && av_lora_dx_masked_supported(dt, d, R, lt, la, nj, d, d)) {
                    AV_TRY(av_lora_dx_masked(Tp, lt, Ap, la, sd, nj, R, w.dxn, d, w.dxn, d, M, d, m->lora_dropout, m->dropout_seed_dev, dt, st));
                } else
                for (int j = 0; j < 3; ++j) {
                    const avllm_lora_mod& lj = P.lora[j];
                    if (!lj.A_pad) continue;
                    avllm_gemm_desc gm = gemm_desc(dt, (char*)w.dtqkv + (size_t)j * AVLLM_LORA_PAD * es, 3 * AVLLM_LORA_PAD, lj.AT_pad, lj.ld_at,
                                                   w.dxn, d, M, d, AVLLM_LORA_PAD);
                    gm.R = w.dxn; gm.ldr = d; gm.drop_seed = m->dropout_seed + 4u * l + j; gm.drop_p = m->lora_dropout; gm.seed_dev = m->dropout_seed_dev;
                    AV_TRY(av_gemm(&gm, st));
                }
            }
            AV_TRY(av_rmsnorm_bwd(w.dxn, resid[l], P.ln1_w, a.rstd1, w.dres, w.dres, M, d, dt, st));
        }
        if (after_layer) after_layer(l, user);
    }
    return AV_OK;
}

// =============================================================================================== Llama inference
namespace {
struct LlamaInferWs { void *x, *xn, *qkv, *att, *t, *gu, *hmid, *logits; float* rope_tab; float* lt; };
void carve_llama_infer(const avllm_llama* m, int B, int S, Bump& b, LlamaInferWs& w, bool all_logits) {
    const size_t es = av_dtype_size(m->dtype);
    const long M = (long)B * S;
    w.x = b.take((size_t)M * m->d * es);
    w.xn = b.take((size_t)M * m->d * es);
    w.qkv = b.take((size_t)M * llama_qw(m) * es);
    w.att = b.take((size_t)M * m->d * es);
    w.t = b.take((size_t)M * AVLLM_LORA_PAD * es);
    w.gu = b.take((size_t)M * 2 * m->ffn * es);
    w.hmid = b.take((size_t)M * m->ffn * es);
    w.logits = b.take((size_t)(all_logits ? M : B) * m->vocab * 4);
    w.rope_tab = (float*)b.take((size_t)S * (m->d / m->heads) * 4);
    w.lt = (float*)b.take((size_t)16 * 4 * AVLLM_LORA_PAD * 4);      // fused token step with adapters: rank-side products [B <= 16, 4 modules x 64] f32
}

// one decoder block on M = B*S rows at positions [pos0, pos0+S); K/V appended to the cache
int llama_infer_layer(const avllm_llama* m, int l, LlamaInferWs& w, int B, int S, int pos0, void* kc, void* vc, int Tmax, hipStream_t st) {
    const avllm_llama_layer& P = m->layer[l];
    const int dt = m->dtype, d = m->d, f = m->ffn, H = m->heads, hd = d / H, M = B * S;
    const int Hkv = llama_kv_heads(m), dkv = llama_dkv(m), qw = llama_qw(m);
    const size_t es = av_dtype_size(dt);
    char* kcl = (char*)kc + (size_t)l * B * Tmax * dkv * es;
    char* vcl = (char*)vc + (size_t)l * B * Tmax * dkv * es;
    AV_TRY(av_rmsnorm_fwd(w.x, P.ln1_w, w.xn, nullptr, M, d, m->eps, dt, st));
    if (!P.lora[0].A_pad && !P.lora[1].A_pad && !P.lora[2].A_pad) {      // no adapters (decode.py path): one fused q|k|v projection
        avllm_gemm_desc gq = gemm_desc(dt, w.xn, d, P.wqkv, d, w.qkv, qw, M, qw, d);
        AV_TRY(av_gemm(&gq, st));
    } else
    for (int j = 0; j < 3; ++j)
        AV_TRY(lora_proj(m, w.xn, d, (const char*)P.wqkv + (size_t)llama_off(m, j) * d * es, d, d, llama_wid(m, j), P.lora[j], w.t, AVLLM_LORA_PAD,
                         (char*)w.qkv + (size_t)llama_off(m, j) * es, qw, nullptr, 0, M, st));
    char* qkv = (char*)w.qkv;
    if (l == 0) AV_TRY(av_rope_table(w.rope_tab, S, hd, pos0, m->theta, st, nullptr, llama_rope_scale(m)));
    AV_TRY(av_rope_tab(qkv, qw, M, S, H + Hkv, hd, w.rope_tab, 0, dt, st));
    AV_TRY(av_kv_append(qkv + (size_t)d * es, qkv + (size_t)(d + dkv) * es, qw, kcl, vcl, B, S, pos0, Tmax, dkv, dt, st));
    const float scale = 1.0f / sqrtf((float)hd);
    if (S == 1) {
        AV_TRY(av_attention_decode(qkv, qw, kcl, vcl, w.att, d, B, H, hd, pos0 + 1, Tmax, scale, dt, st, H / Hkv));
    } else {
        AV_CHECK_ARG(pos0 == 0, "llama prefill must start at position 0");
        AV_TRY(av_attention_fwd(qkv, qkv + (size_t)d * es, qkv + (size_t)(d + dkv) * es, w.att, nullptr, B, S, S, H, hd, qw, qw, qw,
                                d, scale, 1, dt, 0, st, Hkv));
    }
    AV_TRY(lora_proj(m, w.att, d, P.wo, d, d, d, P.lora[3], w.t, AVLLM_LORA_PAD, w.x, d, w.x, d, M, st));
    AV_TRY(av_rmsnorm_fwd(w.x, P.ln2_w, w.xn, nullptr, M, d, m->eps, dt, st));
    avllm_gemm_desc g = gemm_desc(dt, w.xn, d, P.wgu, d, w.gu, 2 * f, M, 2 * f, d);
    AV_TRY(av_gemm(&g, st));
    AV_TRY(av_swiglu_fwd(w.gu, w.hmid, M, f, dt, st));
    g = gemm_desc(dt, w.hmid, f, P.wdown, f, w.x, d, M, d, f);
    g.R = w.x; g.ldr = d;
    return av_gemm(&g, st);
}
}  // namespace

extern "C" size_t avllm_llama_infer_workspace_bytes(const avllm_llama* m, int32_t B, int32_t S) {
    Bump b(nullptr, (size_t)-1);
    LlamaInferWs w;
    carve_llama_infer(m, B, S, b, w, true);
    return bump_size(b);
}

extern "C" int avllm_llama_prefill(const avllm_llama* m, const void* x, int32_t B, int32_t S, void* kcache, void* vcache,
                                   int32_t Tmax, float* logits_last, void* all_logits, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AV_TRY(check_llama(m));
    AV_CHECK_ARG(x && kcache && vcache && ws && B > 0 && S > 0 && S <= Tmax, "llama_prefill: bad args (S=%d Tmax=%d)", S, Tmax);
    Bump b(ws, ws_bytes);
    LlamaInferWs w;
    carve_llama_infer(m, B, S, b, w, true);
    if (!b.ok) return av_set_error(AV_ERR_WORKSPACE, "llama_prefill: workspace %zu < %zu bytes", ws_bytes, bump_size(b));
    const int dt = m->dtype, d = m->d, M = B * S;
    const size_t es = av_dtype_size(dt);
    AV_HIP(hipMemcpyAsync(w.x, x, (size_t)M * d * es, hipMemcpyDeviceToDevice, st));
    for (int l = 0; l < m->layers; ++l) AV_TRY(llama_infer_layer(m, l, w, B, S, 0, kcache, vcache, Tmax, st));
    AV_TRY(av_rmsnorm_fwd(w.x, m->norm_w, w.xn, nullptr, M, d, m->eps, dt, st));
    if (all_logits) {     // eval-mode forward(): logits for every position, in model dtype
        avllm_gemm_desc g = gemm_desc(dt, w.xn, d, m->lm_head, d, all_logits, m->vocab, M, m->vocab, d);
        AV_TRY(av_gemm(&g, st));
    }
    if (logits_last) {    // rows S-1, 2S-1, ... -> [B,vocab] f32
        avllm_gemm_desc g = gemm_desc(dt, (const char*)w.xn + (size_t)(S - 1) * d * es, (long)S * d, m->lm_head, d, logits_last, m->vocab, B, m->vocab, d);
        g.out_f32 = 1;
        AV_TRY(av_gemm(&g, st));
    }
    return AV_OK;
}

// One decoder block of a token step in 5 launches (decode.hip): bf16, B <= 16 sequences; 7 with adapters (rank <= 16: the rank-side
// products of q|k|v and of o are two more -- tiny -- launches over the A images, the B side rides in the projections' epilogues).
static bool llama_decode_lora_ok(const avllm_llama* m, const avllm_llama_layer& P, bool& any) {
    int n = 0;
    for (int j = 0; j < 4; ++j) n += P.lora[j].A_pad != nullptr;
    any = n > 0;
    if (n == 0) return true;
    if (n != 4 || m->lora_r < 1 || m->lora_r > 16) return false;
    for (int j = 0; j < 4; ++j) if (!P.lora[j].B_pad) return false;
    // the A images of q, k, v must form one [3 x 64, d] matrix (avllm/engine.py allocates them that way): one launch makes all three products
    const size_t step = (size_t)AVLLM_LORA_PAD * m->d * 2;
    return (const char*)P.lora[1].A_pad == (const char*)P.lora[0].A_pad + step && (const char*)P.lora[2].A_pad == (const char*)P.lora[0].A_pad + 2 * step;
}
static bool llama_decode_fused_ok(const avllm_llama* m, int B) {
    const bool off = av_knob(AV_KNOB_DECODE_FUSED) == 0;
    if (off || m->dtype != AV_BF16 || B > 16) return false;
    const int hd = m->d / m->heads;
    if (!(hd == 64 || hd == 128)) return false;
    if (!av_dec_proj_supported(AV_BF16, B, m->d, llama_qw(m), 2, hd) || !av_dec_proj_supported(AV_BF16, B, m->d, m->ffn, 1, hd) ||
        !av_dec_proj_supported(AV_BF16, B, m->ffn, m->d, 0, hd) || !av_dec_proj_supported(AV_BF16, B, m->d, m->d, 0, hd)) return false;
    bool any;
    for (int l = 0; l < m->layers; ++l)
        if (!llama_decode_lora_ok(m, m->layer[l], any)) return false;      // adapters the epilogue form does not cover: the general path (lora_proj)
    return true;
}

static int llama_decode_layer_fused(const avllm_llama* m, int l, LlamaInferWs& w, int B, int pos, const int* pos_dev, void* kc, void* vc, int Tmax,
                                    hipStream_t st) {
    const avllm_llama_layer& P = m->layer[l];
    const int d = m->d, f = m->ffn, H = m->heads, hd = d / H, Hkv = llama_kv_heads(m), dkv = llama_dkv(m), qw = llama_qw(m);
    char* kcl = (char*)kc + (size_t)l * B * Tmax * dkv * 2;
    char* vcl = (char*)vc + (size_t)l * B * Tmax * dkv * 2;
    bool lora = false;
    llama_decode_lora_ok(m, P, lora);
    constexpr int LT = 4 * AVLLM_LORA_PAD;
    avllm_dec_proj_desc p = {};
    if (lora) {      // lora_A(rmsnorm(x)) of q, k, v in one launch over the stacked A images [3 x 64, d] -> lt[:, 0:192]
        p.A = w.x; p.lda = d; p.W = P.lora[0].A_pad; p.ldw = d; p.norm_w = P.ln1_w; p.eps = m->eps; p.M = B; p.K = d; p.N = 3 * AVLLM_LORA_PAD; p.mode = 0;
        p.C = w.lt; p.ldc = LT; p.out_f32 = 1;
        AV_TRY(av_dec_proj(&p, st));
        p = {};
    }
    p.A = w.x; p.lda = d; p.W = P.wqkv; p.ldw = d; p.norm_w = P.ln1_w; p.eps = m->eps; p.M = B; p.K = d; p.N = qw; p.mode = 2;
    p.C = w.qkv; p.ldc = qw; p.dq = d; p.dkv = dkv; p.hd = hd; p.rope = w.rope_tab; p.kc = kcl; p.vc = vcl; p.Tmax = Tmax; p.pos = pos; p.pos_dev = pos_dev;
    if (lora) {
        p.lora_t = w.lt; p.ld_lora_t = LT; p.lora_r = m->lora_r; p.lora_scale = m->lora_scale;
        for (int j = 0; j < 3; ++j) p.lora_b[j] = P.lora[j].B_pad;
    }
    AV_TRY(av_dec_proj(&p, st));
    AV_TRY(av_attention_decode1(w.qkv, qw, kcl, vcl, w.att, d, B, H, hd, pos + 1, pos_dev, Tmax, 1.0f / sqrtf((float)hd), AV_BF16, st, H / Hkv));
    p = {};
    if (lora) {      // lora_A(attention output) -> lt[:, 192:192+16]: only the rank's own 16 rows of the padded image are streamed
        p.A = w.att; p.lda = d; p.W = P.lora[3].A_pad; p.ldw = d; p.M = B; p.K = d; p.N = 16; p.mode = 0; p.C = w.lt + 3 * AVLLM_LORA_PAD; p.ldc = LT; p.out_f32 = 1;
        AV_TRY(av_dec_proj(&p, st));
        p = {};
    }
    p.A = w.att; p.lda = d; p.W = P.wo; p.ldw = d; p.M = B; p.K = d; p.N = d; p.mode = 0; p.C = w.x; p.ldc = d; p.R = w.x; p.ldr = d;
    if (lora) { p.lora_t = w.lt + 3 * AVLLM_LORA_PAD; p.ld_lora_t = LT; p.lora_r = m->lora_r; p.lora_scale = m->lora_scale; p.lora_b[0] = P.lora[3].B_pad; }
    AV_TRY(av_dec_proj(&p, st));
    p = {};
    p.A = w.x; p.lda = d; p.W = P.wgu; p.ldw = d; p.norm_w = P.ln2_w; p.eps = m->eps; p.M = B; p.K = d; p.N = f; p.mode = 1; p.C = w.hmid; p.ldc = f;
    AV_TRY(av_dec_proj(&p, st));
    p = {};
    p.A = w.hmid; p.lda = f; p.W = P.wdown; p.ldw = f; p.M = B; p.K = f; p.N = d; p.mode = 0; p.C = w.x; p.ldc = d; p.R = w.x; p.ldr = d;
    return av_dec_proj(&p, st);
}

extern "C" int avllm_llama_decode_step_at(const avllm_llama* m, const int64_t* ids, int32_t B, int32_t pos, const int32_t* pos_dev, void* kcache,
                                          void* vcache, int32_t Tmax, float* logits, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AV_TRY(check_llama(m));
    AV_CHECK_ARG(ids && kcache && vcache && logits && ws && B > 0 && pos >= 0 && (pos_dev || pos < Tmax), "llama_decode_step: bad args (pos=%d Tmax=%d)", pos, Tmax);
    Bump b(ws, ws_bytes);
    LlamaInferWs w;
    carve_llama_infer(m, B, 1, b, w, true);
    if (!b.ok) return av_set_error(AV_ERR_WORKSPACE, "llama_decode_step: workspace %zu < %zu bytes", ws_bytes, bump_size(b));
    const int dt = m->dtype, d = m->d;
    AV_TRY(av_embedding(m->embed, ids, w.x, B, d, dt, st));
    if (llama_decode_fused_ok(m, B)) {
        AV_TRY(av_rope_table(w.rope_tab, 1, d / m->heads, pos, m->theta, st, pos_dev, llama_rope_scale(m)));
        for (int l = 0; l < m->layers; ++l) AV_TRY(llama_decode_layer_fused(m, l, w, B, pos, pos_dev, kcache, vcache, Tmax, st));
        if (av_dec_proj_supported(AV_BF16, B, d, m->vocab, 0, 0)) {          // final norm folded into the lm_head stream
            avllm_dec_proj_desc p = {};
            p.A = w.x; p.lda = d; p.W = m->lm_head; p.ldw = d; p.norm_w = m->norm_w; p.eps = m->eps; p.M = B; p.K = d; p.N = m->vocab; p.mode = 0;
            p.C = logits; p.ldc = m->vocab; p.out_f32 = 1;
            return av_dec_proj(&p, st);
        }
    } else {
        AV_CHECK_ARG(!pos_dev, "llama_decode_step: a device-side position needs the fused bf16 token step (B <= 16, adapters of rank <= 16 or none)");
        for (int l = 0; l < m->layers; ++l) AV_TRY(llama_infer_layer(m, l, w, B, 1, pos, kcache, vcache, Tmax, st));
    }
    AV_TRY(av_rmsnorm_fwd(w.x, m->norm_w, w.xn, nullptr, B, d, m->eps, dt, st));
    avllm_gemm_desc g = gemm_desc(dt, w.xn, d, m->lm_head, d, logits, m->vocab, B, m->vocab, d);
    g.out_f32 = 1;
    return av_gemm(&g, st);
}

extern "C" int avllm_llama_decode_is_fused(const avllm_llama* m, int32_t B) { return m && check_llama(m) == AV_OK && llama_decode_fused_ok(m, B) ? 1 : 0; }

extern "C" int avllm_llama_decode_step(const avllm_llama* m, const int64_t* ids, int32_t B, int32_t pos, void* kcache,
                                       void* vcache, int32_t Tmax, float* logits, void* ws, size_t ws_bytes, void* stream) {
    return avllm_llama_decode_step_at(m, ids, B, pos, nullptr, kcache, vcache, Tmax, logits, ws, ws_bytes, stream);
}
