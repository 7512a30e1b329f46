// placeholder until the MFMA backward lands: route to the scalar kernels
#include "common.h"
#include "avllm_internal.h"
int av_attention_bwd_ref(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                         void* dq, void* dk, void* dv, int B, int T, int H, int hd, long ldq, long ldk, long ldv, long lddo,
                         long lddq, long lddk, long lddv, float scale, int causal, int dtype, hipStream_t st);
int av_attention_bwd_mfma(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                          void* dq, void* dk, void* dv, int B, int T, int H, int hd, long ldq, long ldk, long ldv, long lddo,
                          long lddq, long lddk, long lddv, float scale, int causal, hipStream_t st) {
    return av_attention_bwd_ref(q, k, v, dout, lse, delta, dq, dk, dv, B, T, H, hd, ldq, ldk, ldv, lddo, lddq, lddk, lddv, scale, causal, AV_BF16, st);
}
