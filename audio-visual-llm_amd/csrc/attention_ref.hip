// Scalar (one wave per row) softmax attention, forward and backward, any dtype / head_dim <= 128.
// This is the strict-parity (fp32) implementation and the in-library cross-check for the MFMA kernels
// (impl=1 in avllm_attention_fwd/bwd); it is NOT the bf16 hot path.  Also: delta = rowsum(dO*O) and the
// single-token KV-cache attention used by greedy decode (HBM-bound: K/V rows streamed once, 16-B loads).
// Reference arithmetic: HF eager_attention_forward (softmax(QK^T*scale+mask)V), autograd for the backward.
#include "common.h"
#include "avllm_internal.h"

namespace {

constexpr int MAXHD = 128;

template <typename T>
__device__ __forceinline__ float dot_row(const float* __restrict__ qs, const T* __restrict__ kr, int hd) {
    float s = 0.f;
    for (int c = 0; c < hd; c += 8) {
        float kv[8];
        load_f<8>(kr + c, kv);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += qs[c + j] * kv[j];
    }
    return s;
}

// grid (ceil(Tq/4), H, B); wave w -> query t.  NO = 64-wide chunks of the head dim a lane accumulates: 2 (head_dim <= 128: every model of the
// path) or 8 (<= 512: the modality connectors' nn.MultiheadAttention, 8 heads over the LLM width -- modality_connector.py:186-191, 347-352)
template <typename T, int NO>
__global__ __launch_bounds__(256) void attn_fwd_ref(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                    T* __restrict__ o, float* __restrict__ lse, int Tq, int Tk, int H, int hd,
                                                    long ldq, long ldk, long ldv, long ldo, float scale, int causal, int G) {
    __shared__ float qs[4][NO * 64];
    __shared__ float ps[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
    if (t >= Tq) return;
    const T* qr = q + ((long)b * Tq + t) * ldq + (long)h * hd;
    for (int c = lane; c < hd; c += 64) qs[w][c] = to_f(qr[c]);
    __builtin_amdgcn_wave_barrier();
    const int limit = causal ? t + (Tk - Tq) : Tk - 1;      // last visible key
    float m = -INFINITY, l = 0.f, oacc[NO];
#pragma unroll
    for (int x = 0; x < NO; ++x) oacc[x] = 0.f;
    for (int kb = 0; kb <= limit && kb < Tk; kb += 64) {
        const int j = kb + lane;
        const bool valid = j < Tk && j <= limit;
        float s = -INFINITY;
        if (valid) s = scale * dot_row(qs[w], k + ((long)b * Tk + j) * ldk + (long)(h / G) * hd, hd);
        const float mn = fmaxf(m, wave_max(s));
        const float p = valid ? __expf(s - mn) : 0.f;
        const float alpha = __expf(m - mn);
        l = l * alpha + wave_sum(p);
        m = mn;
        __builtin_amdgcn_wave_barrier();
        ps[w][lane] = p;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int x = 0; x < NO; ++x) oacc[x] *= alpha;
        const int nj = min(64, min(Tk, limit + 1) - kb);
        for (int jj = 0; jj < nj; ++jj) {
            const T* vr = v + ((long)b * Tk + kb + jj) * ldv + (long)(h / G) * hd;
            const float pj = ps[w][jj];
#pragma unroll
            for (int x = 0; x < NO; ++x)
                if (lane + 64 * x < hd) oacc[x] += pj * to_f(vr[lane + 64 * x]);
        }
    }
    T* orow = o + ((long)b * Tq + t) * ldo + (long)h * hd;
    const float inv = 1.0f / l;
#pragma unroll
    for (int x = 0; x < NO; ++x)
        if (lane + 64 * x < hd) orow[lane + 64 * x] = from_f<T>(oacc[x] * inv);
    if (lse && lane == 0) lse[((long)b * H + h) * Tq + t] = m + logf(l);
}

// delta[b,h,t] = sum_d dO*O ; grid (ceil(T/4), H, B)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ delta,
                                                         int Tn, int H, int hd, long ldo, long lddo) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
    if (t >= Tn) return;
    const T* orow = o + ((long)b * Tn + t) * ldo + (long)h * hd;
    const T* drow = dout + ((long)b * Tn + t) * lddo + (long)h * hd;
    float s = 0.f;
    for (int c = lane; c < hd; c += 64) s += to_f(orow[c]) * to_f(drow[c]);
    s = wave_sum(s);
    if (lane == 0) delta[((long)b * H + h) * Tn + t] = s;
}

// dQ: wave per query
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_ref(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                       const T* __restrict__ dout, const float* __restrict__ lse,
                                                       const float* __restrict__ delta, T* __restrict__ dq, int Tn, int H, int hd,
                                                       long ldq, long ldk, long ldv, long lddo, long lddq, float scale, int causal, int G) {
    __shared__ float qs[4][MAXHD], dos[4][MAXHD];
    __shared__ float dss[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
    if (t >= Tn) return;
    const T* qr = q + ((long)b * Tn + t) * ldq + (long)h * hd;
    const T* dr = dout + ((long)b * Tn + t) * lddo + (long)h * hd;
    for (int c = lane; c < hd; c += 64) { qs[w][c] = to_f(qr[c]); dos[w][c] = to_f(dr[c]); }
    __builtin_amdgcn_wave_barrier();
    const float L = lse[((long)b * H + h) * Tn + t], D = delta[((long)b * H + h) * Tn + t];
    const int limit = causal ? t : Tn - 1;
    float g0 = 0.f, g1 = 0.f;
    for (int kb = 0; kb <= limit; kb += 64) {
        const int j = kb + lane;
        const bool valid = j <= limit;
        float ds = 0.f;
        if (valid) {
            const float s = scale * dot_row(qs[w], k + ((long)b * Tn + j) * ldk + (long)(h / G) * hd, hd);
            const float p = __expf(s - L);
            const float dp = dot_row(dos[w], v + ((long)b * Tn + j) * ldv + (long)(h / G) * hd, hd);
            ds = p * (dp - D) * scale;
        }
        __builtin_amdgcn_wave_barrier();
        dss[w][lane] = ds;
        __builtin_amdgcn_wave_barrier();
        const int nj = min(64, limit + 1 - kb);
        for (int jj = 0; jj < nj; ++jj) {
            const T* kr = k + ((long)b * Tn + kb + jj) * ldk + (long)(h / G) * hd;
            const float d = dss[w][jj];
            if (lane < hd) g0 += d * to_f(kr[lane]);
            if (lane + 64 < hd) g1 += d * to_f(kr[lane + 64]);
        }
    }
    T* out = dq + ((long)b * Tn + t) * lddq + (long)h * hd;
    if (lane < hd) out[lane] = from_f<T>(g0);
    if (lane + 64 < hd) out[lane + 64] = from_f<T>(g1);
}

// dK,dV: wave per key
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_ref(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        const T* __restrict__ dout, const float* __restrict__ lse,
                                                        const float* __restrict__ delta, T* __restrict__ dk, T* __restrict__ dv, int Tn,
                                                        int H, int hd, long ldq, long ldk, long ldv, long lddo, long lddk, long lddv,
                                                        float scale, int causal, int G) {
    __shared__ float ks[4][MAXHD], vs[4][MAXHD];
    __shared__ float ps[4][64], dss[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
    if (j >= Tn) return;
    const T* kr = k + ((long)b * Tn + j) * ldk + (long)h * hd;
    const T* vr = v + ((long)b * Tn + j) * ldv + (long)h * hd;
    for (int c = lane; c < hd; c += 64) { ks[w][c] = to_f(kr[c]); vs[w][c] = to_f(vr[c]); }
    __builtin_amdgcn_wave_barrier();
    const int first = causal ? j : 0;
    float gk0 = 0.f, gk1 = 0.f, gv0 = 0.f, gv1 = 0.f;
    for (int g = 0; g < G; ++g) {                          // h is the K/V head; its G query heads all contribute
    const int hq = h * G + g;
    for (int qb = (first / 64) * 64; qb < Tn; qb += 64) {
        const int i = qb + lane;
        const bool valid = i < Tn && i >= first;
        float p = 0.f, ds = 0.f;
        if (valid) {
            const float s = scale * dot_row(ks[w], q + ((long)b * Tn + i) * ldq + (long)hq * hd, hd);
            p = __expf(s - lse[((long)b * H + hq) * Tn + i]);
            const float dp = dot_row(vs[w], dout + ((long)b * Tn + i) * lddo + (long)hq * hd, hd);
            ds = p * (dp - delta[((long)b * H + hq) * Tn + i]) * scale;
        }
        __builtin_amdgcn_wave_barrier();
        ps[w][lane] = p; dss[w][lane] = ds;
        __builtin_amdgcn_wave_barrier();
        const int ni = min(64, Tn - qb);
        for (int ii = 0; ii < ni; ++ii) {
            const T* qr = q + ((long)b * Tn + qb + ii) * ldq + (long)hq * hd;
            const T* dr = dout + ((long)b * Tn + qb + ii) * lddo + (long)hq * hd;
            const float pp = ps[w][ii], dd = dss[w][ii];
            if (lane < hd) { gv0 += pp * to_f(dr[lane]); gk0 += dd * to_f(qr[lane]); }
            if (lane + 64 < hd) { gv1 += pp * to_f(dr[lane + 64]); gk1 += dd * to_f(qr[lane + 64]); }
        }
    }
    }
    T* ok = dk + ((long)b * Tn + j) * lddk + (long)h * hd;
    T* ov = dv + ((long)b * Tn + j) * lddv + (long)h * hd;
    if (lane < hd) { ok[lane] = from_f<T>(gk0); ov[lane] = from_f<T>(gv0); }
    if (lane + 64 < hd) { ok[lane + 64] = from_f<T>(gk1); ov[lane + 64] = from_f<T>(gv1); }
}

// single-token attention over a KV cache [B,Tmax,d]: block per (b,h), 4 waves.  HBM/L2-bound row streaming: a group of
// G = hd/8 lanes covers one K (or V) row with 16-byte loads, so one wave-instruction reads 64/G whole rows, coalesced.
template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ kc,
                                                          const T* __restrict__ vc, T* __restrict__ o, long ldo, int H, int hd, int Tk,
                                                          int Tmax, float scale, int GQ) {
    extern __shared__ float sh[];            // [Tk] scores | 8 reduction | [4][hd] partial outputs
    float* sc = sh; float* red = sc + Tk; float* part = red + 8;
    const int h = blockIdx.x, b = blockIdx.y, d = (H / GQ) * hd;      // cache rows hold the H/GQ key/value heads
    const int G = hd >> 3;                   // lanes per row (16 for hd=128, 8 for hd=64)
    const int rows_per_pass = 256 / G;
    const int tid = threadIdx.x, dc = tid % G, rsub = tid / G;
    float qv[8];
    load_f<8>(q + (long)b * ldq + (long)h * hd + dc * 8, qv);
    const T* kbase = kc + ((long)b * Tmax) * d + (long)(h / GQ) * hd + dc * 8;
    const T* vbase = vc + ((long)b * Tmax) * d + (long)(h / GQ) * hd + dc * 8;
    float mx = -INFINITY;
    // four row groups per trip, all four loads issued before the first is used: the pass is a chain of memory latencies otherwise
    // (one 16-byte load per thread in flight: 30 us per layer at Tk = 300 against ~8 us of K/V bytes)
    constexpr int UN = 4;
    for (int t0 = 0; t0 < Tk; t0 += UN * rows_per_pass) {
        float kv[UN][8];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int t = t0 + u * rows_per_pass + rsub;
#pragma unroll
            for (int j = 0; j < 8; ++j) kv[u][j] = 0.f;
            if (t < Tk) load_f<8>(kbase + (long)t * d, kv[u]);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int t = t0 + u * rows_per_pass + rsub;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += qv[j] * kv[u][j];
            for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off);      // reduce inside the G-lane group
            s *= scale;
            if (t < Tk) { if (dc == 0) sc[t] = s; mx = fmaxf(mx, s); }
        }
    }
    mx = block_max(mx, red);
    float sm = 0.f;
    for (int j = tid; j < Tk; j += 256) { const float p = __expf(sc[j] - mx); sc[j] = p; sm += p; }
    sm = block_sum(sm, red);                 // (contains the barriers that publish sc[])
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < Tk; t0 += UN * rows_per_pass) {
        float vv[UN][8], p[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int t = t0 + u * rows_per_pass + rsub;
#pragma unroll
            for (int j = 0; j < 8; ++j) vv[u][j] = 0.f;
            p[u] = 0.f;
            if (t < Tk) { load_f<8>(vbase + (long)t * d, vv[u]); p[u] = sc[t]; }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += p[u] * vv[u][j];
    }
    // reduce over the row groups of a wave (lanes with equal dc), then over the 4 waves through LDS
#pragma unroll
    for (int j = 0; j < 8; ++j)
        for (int off = G; off < 64; off <<= 1) acc[j] += __shfl_xor(acc[j], off);
    const int w = tid >> 6, lane = tid & 63;
    if (lane < G) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[w * hd + lane * 8 + j] = acc[j];
    }
    __syncthreads();
    if (tid < hd) {
        const float t = part[tid] + part[hd + tid] + part[2 * hd + tid] + part[3 * hd + tid];
        o[(long)b * ldo + (long)h * hd + tid] = from_f<T>(t / sm);
    }
}

}  // namespace

int av_attention_fwd_ref(const void* q, const void* k, const void* v, void* o, float* lse, int B, int Tq, int Tk, int H,
                         int hd, long ldq, long ldk, long ldv, long ldo, float scale, int causal, int dtype, hipStream_t st, int G) {
    AV_CHECK_ARG(hd % 8 == 0 && hd <= 4 * MAXHD, "attention(ref): head_dim %d unsupported (multiple of 8, <= 512)", hd);
    const dim3 grid(av_cdiv(Tq, 4), H, B);
#define AV_FWD_REF(TT, NOV) hipLaunchKernelGGL((attn_fwd_ref<TT, NOV>), grid, dim3(256), 0, st, (const TT*)q, (const TT*)k, (const TT*)v, (TT*)o, lse, Tq, Tk, H, hd, ldq, ldk, ldv, ldo, scale, causal, G)
    if (hd <= MAXHD) { if (dtype == AV_F32) AV_FWD_REF(float, 2); else AV_FWD_REF(bf16, 2); }
    else { if (dtype == AV_F32) AV_FWD_REF(float, 8); else AV_FWD_REF(bf16, 8); }
#undef AV_FWD_REF
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_attention_delta(const void* o, const void* dout, float* delta, int B, int T, int H, int hd, long ldo, long lddo, int dtype, hipStream_t st) {
    const dim3 grid(av_cdiv(T, 4), H, B);
    if (dtype == AV_F32) hipLaunchKernelGGL((attn_delta_kernel<float>), grid, dim3(256), 0, st, (const float*)o, (const float*)dout, delta, T, H, hd, ldo, lddo);
    else hipLaunchKernelGGL((attn_delta_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)o, (const bf16*)dout, delta, T, H, hd, ldo, lddo);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_attention_bwd_ref(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                         void* dq, void* dk, void* dv, int B, int T, int H, int hd, long ldq, long ldk, long ldv, long lddo,
                         long lddq, long lddk, long lddv, float scale, int causal, int dtype, hipStream_t st, int G) {
    AV_CHECK_ARG(hd % 8 == 0 && hd <= MAXHD, "attention_bwd(ref): head_dim %d unsupported", hd);
    const dim3 grid(av_cdiv(T, 4), H, B), gridkv(av_cdiv(T, 4), H / G, B);
    if (dtype == AV_F32) {
        hipLaunchKernelGGL((attn_bwd_dq_ref<float>), grid, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse, delta, (float*)dq, T, H, hd, ldq, ldk, ldv, lddo, lddq, scale, causal, G);
        hipLaunchKernelGGL((attn_bwd_dkv_ref<float>), gridkv, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse, delta, (float*)dk, (float*)dv, T, H, hd, ldq, ldk, ldv, lddo, lddk, lddv, scale, causal, G);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_ref<bf16>), grid, dim3(256), 0, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)dout, lse, delta, (bf16*)dq, T, H, hd, ldq, ldk, ldv, lddo, lddq, scale, causal, G);
        hipLaunchKernelGGL((attn_bwd_dkv_ref<bf16>), gridkv, dim3(256), 0, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)dout, lse, delta, (bf16*)dk, (bf16*)dv, T, H, hd, ldq, ldk, ldv, lddo, lddk, lddv, scale, causal, G);
    }
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_attention_decode(const void* q, long ldq, const void* kc, const void* vc, void* o, long ldo, int B, int H, int hd,
                        int Tk, int Tmax, float scale, int dtype, hipStream_t st, int G) {
    AV_CHECK_ARG(q && kc && vc && o && Tk > 0 && Tk <= Tmax && G > 0 && H % G == 0, "attention_decode: bad args");
    AV_CHECK_ARG((hd == 64 || hd == 128) && ldq % 8 == 0, "attention_decode: head_dim %d unsupported", hd);
    const size_t sh = (size_t)(Tk + 8 + 4 * hd) * sizeof(float);
    AV_CHECK_ARG(sh <= 64 * 1024, "attention_decode: Tk=%d too long for the LDS score buffer", Tk);
    const dim3 grid(H, B);
    if (dtype == AV_F32) hipLaunchKernelGGL((attn_decode_kernel<float>), grid, dim3(256), sh, st, (const float*)q, ldq, (const float*)kc, (const float*)vc, (float*)o, ldo, H, hd, Tk, Tmax, scale, G);
    else hipLaunchKernelGGL((attn_decode_kernel<bf16>), grid, dim3(256), sh, st, (const bf16*)q, ldq, (const bf16*)kc, (const bf16*)vc, (bf16*)o, ldo, H, hd, Tk, Tmax, scale, G);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
