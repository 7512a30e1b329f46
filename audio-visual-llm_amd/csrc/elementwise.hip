// HBM-bound byte movers of the hot path: RoPE, SwiGLU, embedding gather, casts, Whisper/CLIP im2col,
// the encode()/forward() fusion + pooling glue, LoRA operand packing, KV-cache append.
// All accesses are 8/16-byte vectors on the contiguous axis; no GEMM-shaped work lives here.
#include "common.h"
#include "avllm_internal.h"

namespace {

// ---------------------------------------------------------------- RoPE (HF llama :129-160, rotate_half form)
template <typename T>
__global__ void rope_kernel(T* __restrict__ x, long ld, long rows, int T_, int heads, int hd, int pos0, float theta, int inverse) {
    const int half = hd >> 1;
    const long total = rows * heads * (half >> 2);          // 4 pairs per thread
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int qd = (int)(idx % (half >> 2));
        const int h = (int)((idx / (half >> 2)) % heads);
        const long row = idx / ((long)(half >> 2) * heads);
        const int pos = pos0 + (int)(row % T_);
        T* p = x + row * ld + (long)h * hd + qd * 4;
        float a[4], b[4], oa[4], ob[4];
        load_f<4>(p, a);
        load_f<4>(p + half, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = qd * 4 + j;
            const float inv = 1.0f / powf(theta, (float)(2 * i) / (float)hd);
            const float ang = (float)pos * inv;
            float c = cosf(ang), s = sinf(ang);
            if (inverse) s = -s;
            oa[j] = a[j] * c - b[j] * s;      // x1*cos - x2*sin
            ob[j] = b[j] * c + a[j] * s;      // x2*cos + x1*sin
        }
        store_f<4>(p, oa);
        store_f<4>(p + half, ob);
    }
}

// cos/sin table [T][hd/2][2] for positions pos0..pos0+T-1 (computed once per call instead of per element per layer)
// sc.orig_ctx > 0: Llama-3.1 / 3.2 "llama3" frequency scaling (HF:modeling_rope_utils.py _compute_llama3_parameters): wavelengths beyond
// orig_ctx / low_freq_factor are stretched by `factor`, those below orig_ctx / high_freq_factor kept, the band between interpolated.
__global__ void rope_table_kernel(float* __restrict__ tab, int T_, int hd, int pos0, float theta, const int* __restrict__ pos_dev, AvRopeScale sc) {
    const int half = hd >> 1;
    if (pos_dev) pos0 += *pos_dev;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T_ * half) return;
    const int i = idx % half, t = idx / half;
    float inv = 1.0f / powf(theta, (float)(2 * i) / (float)hd);
    if (sc.orig_ctx > 0) {
        const float wavelen = 6.283185307179586f / inv, octx = (float)sc.orig_ctx;
        const float low_wl = octx / sc.low_freq_factor, high_wl = octx / sc.high_freq_factor;
        if (wavelen > low_wl) inv = inv / sc.factor;
        else if (!(wavelen < high_wl)) {
            const float smooth = (octx / wavelen - sc.low_freq_factor) / (sc.high_freq_factor - sc.low_freq_factor);
            inv = (1.0f - smooth) * inv / sc.factor + smooth * inv;
        }
    }
    const float ang = (float)(pos0 + t) * inv;
    tab[2 * idx] = cosf(ang);
    tab[2 * idx + 1] = sinf(ang);
}

template <typename T>
__global__ void rope_tab_kernel(T* __restrict__ x, long ld, long rows, int T_, int heads, int hd, const float* __restrict__ tab, int inverse) {
    const int half = hd >> 1;
    const long total = rows * heads * (half >> 2);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int qd = (int)(idx % (half >> 2));
        const int h = (int)((idx / (half >> 2)) % heads);
        const long row = idx / ((long)(half >> 2) * heads);
        const int t = (int)(row % T_);
        T* p = x + row * ld + (long)h * hd + qd * 4;
        float a[4], b[4], oa[4], ob[4], cs[8];
        load_f<4>(p, a);
        load_f<4>(p + half, b);
        load_f<8>(tab + ((long)t * half + qd * 4) * 2, cs);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float c = cs[2 * j], s = inverse ? -cs[2 * j + 1] : cs[2 * j + 1];
            oa[j] = a[j] * c - b[j] * s;
            ob[j] = b[j] * c + a[j] * s;
        }
        store_f<4>(p, oa);
        store_f<4>(p + half, ob);
    }
}

// ---------------------------------------------------------------- SwiGLU
template <typename T>
__global__ void swiglu_fwd_kernel(const T* __restrict__ gu, T* __restrict__ h, long M, int F) {
    const long total = M * (F >> 2);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / (F >> 2);
        const int c = (int)(idx % (F >> 2)) * 4;
        float g[4], u[4], o[4];
        load_f<4>(gu + m * 2 * F + c, g);
        load_f<4>(gu + m * 2 * F + F + c, u);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = g[j] / (1.0f + __expf(-g[j])) * u[j];
        store_f<4>(h + m * F + c, o);
    }
}
template <typename T>
__global__ void swiglu_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ gu, T* __restrict__ dgu, long M, int F) {
    const long total = M * (F >> 2);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / (F >> 2);
        const int c = (int)(idx % (F >> 2)) * 4;
        float g[4], u[4], d[4], dg[4], du[4];
        load_f<4>(gu + m * 2 * F + c, g);
        load_f<4>(gu + m * 2 * F + F + c, u);
        load_f<4>(dh + m * F + c, d);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = 1.0f / (1.0f + __expf(-g[j]));
            const float silu = g[j] * sg;
            du[j] = d[j] * silu;
            dg[j] = d[j] * u[j] * (sg * (1.0f + g[j] * (1.0f - sg)));
        }
        store_f<4>(dgu + m * 2 * F + c, dg);
        store_f<4>(dgu + m * 2 * F + F + c, du);
    }
}

// ---------------------------------------------------------------- y = act(x) + r  (DeepModalityConnector: Linear -> LayerNorm -> GELU [+ residual],
// src/clip_whisper/models/modality_connector.py:91-108: the norm sits between the projection and the activation, so the GEMM epilogue cannot carry it)
template <typename T>
__global__ void act_residual_kernel(const T* __restrict__ x, const T* __restrict__ r, T* __restrict__ y, long n4, int act) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n4; idx += (long)gridDim.x * blockDim.x) {
        float v[4], rr[4] = {0.f, 0.f, 0.f, 0.f};
        load_f<4>(x + idx * 4, v);
        if (r) load_f<4>(r + idx * 4, rr);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], act) + rr[j];
        store_f<4>(y + idx * 4, v);
    }
}

// ---------------------------------------------------------------- dropout (counter-based mask, common.h av_keep)
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int d, uint32_t seed_off, float p, const uint32_t* seed_dev) {
    const uint32_t seed = av_seed(seed_dev, seed_off);
    const long total = rows * (d >> 2);
    const float sc = av_drop_scale(p);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        float v[4];
        load_f<4>(x + idx * 4, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = av_keep(seed, (unsigned long long)(idx * 4 + j), p) ? v[j] * sc : 0.f;
        store_f<4>(y + idx * 4, v);
    }
}

// ---------------------------------------------------------------- embedding / cast
template <typename T>
__global__ void embedding_kernel(const T* __restrict__ table, const int64_t* __restrict__ ids, T* __restrict__ out, long n, int d) {
    const long total = n * (d >> 2);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / (d >> 2);
        const int c = (int)(idx % (d >> 2)) * 4;
        float v[4];
        load_f<4>(table + ids[r] * d + c, v);
        store_f<4>(out + r * d + c, v);
    }
}
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ s, D* __restrict__ d, long n) {
    const long n4 = n >> 2;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n4; idx += (long)gridDim.x * blockDim.x) {
        float v[4];
        load_f<4>(s + idx * 4, v);
        store_f<4>(d + idx * 4, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const long i = (n4 << 2) + threadIdx.x; d[i] = from_f<D>(to_f(s[i])); }
}

// ---------------------------------------------------------------- Whisper conv stem im2col
// conv1: block = 64 consecutive t of one batch item; mel slab [n_mels][66] staged in LDS (coalesced along t)
template <typename T>
__global__ __launch_bounds__(256) void im2col1_kernel(const float* __restrict__ mel, T* __restrict__ cols, int n_mels, int Tn, int Kpad) {
    extern __shared__ float slab[];            // [n_mels][68]
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < n_mels * 66; i += 256) {
        const int c = i / 66, j = i % 66, t = t0 + j - 1;
        slab[c * 68 + j] = (t >= 0 && t < Tn) ? mel[((long)b * n_mels + c) * Tn + t] : 0.f;
    }
    __syncthreads();
    const int K = n_mels * 3;
    for (int i = threadIdx.x; i < 64 * (Kpad >> 2); i += 256) {
        const int tl = i / (Kpad >> 2), col = (i % (Kpad >> 2)) * 4;
        if (t0 + tl >= Tn) continue;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int cc = col + j; o[j] = cc < K ? slab[(cc / 3) * 68 + tl + (cc % 3)] : 0.f; }
        store_f<4>(cols + ((long)b * Tn + t0 + tl) * Kpad + col, o);
    }
}
// conv2 (k3,s2,p1): cols[(b,t'), kw*d + c] = h[b, 2t'+kw-1, c]
template <typename T>
__global__ void im2col2_kernel(const T* __restrict__ h, T* __restrict__ cols, int B, int Tn, int d) {
    const int To = Tn / 2;
    const long total = (long)B * To * 3 * (d >> 3);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % (d >> 3)) * 8;
        const int kw = (int)((idx / (d >> 3)) % 3);
        const long row = idx / ((long)(d >> 3) * 3);
        const int b = (int)(row / To), tp = (int)(row % To);
        const int t = 2 * tp + kw - 1;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < Tn) load_f<8>(h + ((long)b * Tn + t) * d + c, v);
        store_f<8>(cols + row * 3 * d + (long)kw * d + c, v);
    }
}

// ---------------------------------------------------------------- CLIP patchify: coalesced along frame rows
template <typename T, int V, typename TI>         // V pixels per thread: 4 when the patch width allows it, 2 for patch 14 (ViT-L/14), else 1; TI = frame dtype
__global__ void patchify_kernel(const TI* __restrict__ fr, T* __restrict__ cols, int N, int S, int p, int Kpad) {
    const int g = S / p, pp = p * p, SV = S / V;
    const long total = (long)N * 3 * S * SV;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % SV) * V;
        const int y = (int)((idx / SV) % S);
        const int c = (int)((idx / ((long)SV * S)) % 3);
        const long n = idx / ((long)SV * S * 3);
        const TI* src = fr + ((n * 3 + c) * S + y) * S + x;
        const int py = y / p, ky = y % p, px = x / p, kx = x % p;
        T* dst = cols + (n * g * g + (long)py * g + px) * Kpad + c * pp + ky * p + kx;
        if constexpr (V == 4) {
            float o[4];
            load_f<4>(src, o);
            store_f<4>(dst, o);
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i) dst[i] = from_f<T>(to_f(src[i]));
        }
    }
}
template <typename T>
__global__ void patchify_pad_kernel(T* __restrict__ cols, long rows, int K, int Kpad) {
    const int w = Kpad - K;
    const long total = rows * w;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x)
        cols[(idx / w) * Kpad + K + idx % w] = from_f<T>(0.f);
}
template <typename T>
__global__ void cls_rows_kernel(const T* __restrict__ ce, const T* __restrict__ pos, T* __restrict__ x, int N, int tokens, int d) {
    const long total = (long)N * (d >> 2);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long n = idx / (d >> 2);
        const int c = (int)(idx % (d >> 2)) * 4;
        float a[4], b[4];
        load_f<4>(ce + c, a);
        load_f<4>(pos + c, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += b[j];
        store_f<4>(x + n * tokens * d + c, a);
    }
}

// ---------------------------------------------------------------- fusion + adaptive pooling glue
template <typename T>
__device__ __forceinline__ void virt_row(const T* a, int Ta, const T* v, int Tv, const T* pe, int P, int L, int b, int j, int c,
                                         int D, float fs, float (&o)[4]) {
    o[0] = o[1] = o[2] = o[3] = 0.f;
    if (j < P) { load_f<4>(pe + ((long)b * P + j) * D + c, o); return; }
    const int t = j - P;
    if (a && v) {
        float t4[4];
        if (t < Ta) { load_f<4>(a + ((long)b * Ta + t) * D + c, t4);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fs * t4[k]; }
        if (t < Tv) { load_f<4>(v + ((long)b * Tv + t) * D + c, t4);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] += (1.0f - fs) * t4[k]; }
    } else if (a) { if (t < Ta) load_f<4>(a + ((long)b * Ta + t) * D + c, o); }
    else { if (t < Tv) load_f<4>(v + ((long)b * Tv + t) * D + c, o); }
}

template <typename T>
__global__ void fuse_pool_kernel(const T* __restrict__ a, int Ta, const T* __restrict__ v, int Tv, const T* __restrict__ pe,
                                 int P, T* __restrict__ out, int L, int S_out, int D, float fs) {
    const int b = blockIdx.y, i = blockIdx.x;
    const int Lt = P + L;
    for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, r[4];
        if (Lt == S_out) {
            virt_row(a, Ta, v, Tv, pe, P, L, b, i, c, D, fs, acc);
        } else if (Lt > S_out) {          // AdaptiveAvgPool1d window
            const int s = (int)(((long)i * Lt) / S_out);
            const int e = (int)((((long)(i + 1)) * Lt + S_out - 1) / S_out);
            for (int j = s; j < e; ++j) {
                virt_row(a, Ta, v, Tv, pe, P, L, b, j, c, D, fs, r);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += r[k];
            }
            const float cnt = (float)(e - s);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = acc[k] / cnt;
        } else {                          // linear, align_corners=True
            const float scale = S_out > 1 ? (float)(Lt - 1) / (float)(S_out - 1) : 0.f;
            const float src = scale * i;
            const int lo = (int)src;
            const int hi = lo + 1 < Lt ? lo + 1 : Lt - 1;
            const float w1 = src - (float)lo, w0 = 1.0f - w1;
            float r2[4];
            virt_row(a, Ta, v, Tv, pe, P, L, b, lo, c, D, fs, r);
            virt_row(a, Ta, v, Tv, pe, P, L, b, hi, c, D, fs, r2);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = w0 * r[k] + w1 * r2[k];
        }
        store_f<4>(out + ((long)b * S_out + i) * D + c, acc);
    }
}

// ---------------------------------------------------------------- LoRA operand packing
template <typename T>
__global__ void lora_pack_kernel(const float* __restrict__ A, const float* __restrict__ Bm, int r, int din, int dout,
                                 T* __restrict__ A_pad, T* __restrict__ AT_pad, long ld_at, T* __restrict__ B_pad, T* __restrict__ BT_pad) {
    const long nA = (long)AVLLM_LORA_PAD * din, nB = (long)AVLLM_LORA_PAD * dout;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < nA + nB; idx += (long)gridDim.x * blockDim.x) {
        if (idx < nA) {
            const int j = (int)(idx / din), k = (int)(idx % din);
            const float val = j < r ? A[(long)j * din + k] : 0.f;
            A_pad[(long)j * din + k] = from_f<T>(val);
            AT_pad[(long)k * ld_at + j] = from_f<T>(val);
        } else {
            const long i2 = idx - nA;
            const int j = (int)(i2 / dout), n = (int)(i2 % dout);
            const float val = j < r ? Bm[(long)n * r + j] : 0.f;
            BT_pad[(long)j * dout + n] = from_f<T>(val);
            B_pad[(long)n * AVLLM_LORA_PAD + j] = from_f<T>(val);
        }
    }
}

// every adapter of the model in ONE launch: blockIdx.x = module (table row in device memory), blockIdx.y strides over its elements
template <typename T>
__global__ void lora_pack_batch_kernel(const avllm_lora_pack_item* __restrict__ items, int r, int din) {
    const avllm_lora_pack_item it = items[blockIdx.x];
    const int dout = (int)it.dout;
    // only the r real rank rows/columns are rewritten: the padding up to AVLLM_LORA_PAD is zero from allocation and never changes
    const long nA = (long)r * din, nB = (long)r * dout;
    T* A_pad = (T*)it.A_pad; T* AT_pad = (T*)it.AT_pad; T* B_pad = (T*)it.B_pad; T* BT_pad = (T*)it.BT_pad;
    for (long idx = blockIdx.y * (long)blockDim.x + threadIdx.x; idx < nA + nB; idx += (long)gridDim.y * blockDim.x) {
        if (idx < nA) {
            const int j = (int)(idx / din), k = (int)(idx % din);
            const float val = it.A[(long)j * din + k];
            A_pad[(long)j * din + k] = from_f<T>(val);
            AT_pad[(long)k * it.ld_at + j] = from_f<T>(val);
        } else {
            const long i2 = idx - nA;
            const int j = (int)(i2 / dout), n = (int)(i2 % dout);
            const float val = it.B[(long)n * r + j];
            BT_pad[(long)j * dout + n] = from_f<T>(val);
            B_pad[(long)n * AVLLM_LORA_PAD + j] = from_f<T>(val);
        }
    }
}

// ---------------------------------------------------------------- KV cache append: cache[b, pos0+t, :] = k[b*T+t, :]
template <typename T>
__global__ void kv_append_kernel(const T* __restrict__ k, const T* __restrict__ v, long ld, T* __restrict__ kc, T* __restrict__ vc,
                                 int B, int Tn, int pos0, int Tmax, int d) {
    const long total = (long)B * Tn * (d >> 2);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % (d >> 2)) * 4;
        const long row = idx / (d >> 2);
        const int b = (int)(row / Tn), t = (int)(row % Tn);
        float x[4];
        load_f<4>(k + row * ld + c, x);
        store_f<4>(kc + ((long)b * Tmax + pos0 + t) * d + c, x);
        load_f<4>(v + row * ld + c, x);
        store_f<4>(vc + ((long)b * Tmax + pos0 + t) * d + c, x);
    }
}

inline int grid_for(long total, int block = 256) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

int av_rope(void* x, long ld, long rows, int T, int heads, int hd, int pos0, float theta, int inverse, int dtype, hipStream_t st) {
    AV_CHECK_ARG(x && rows > 0 && T > 0, "rope: null/empty");
    AV_CHECK_ARG(hd % 8 == 0 && ld % 4 == 0, "rope: head_dim %d must be a multiple of 8", hd);
    const long total = rows * heads * (hd / 8);
    if (dtype == AV_F32) hipLaunchKernelGGL((rope_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (float*)x, ld, rows, T, heads, hd, pos0, theta, inverse);
    else hipLaunchKernelGGL((rope_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, st, (bf16*)x, ld, rows, T, heads, hd, pos0, theta, inverse);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_rope_table(float* tab, int T, int hd, int pos0, float theta, hipStream_t st, const int* pos_dev, AvRopeScale sc) {
    AV_CHECK_ARG(tab && T > 0 && hd % 8 == 0, "rope_table: bad args");
    hipLaunchKernelGGL(rope_table_kernel, dim3(av_cdiv((long)T * (hd / 2), 256)), dim3(256), 0, st, tab, T, hd, pos0, theta, pos_dev, sc);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_rope_tab(void* x, long ld, long rows, int T, int heads, int hd, const float* tab, int inverse, int dtype, hipStream_t st) {
    AV_CHECK_ARG(x && tab && rows > 0 && T > 0 && hd % 8 == 0 && ld % 4 == 0, "rope(tab): bad args");
    const long total = rows * heads * (hd / 8);
    if (dtype == AV_F32) hipLaunchKernelGGL((rope_tab_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (float*)x, ld, rows, T, heads, hd, tab, inverse);
    else hipLaunchKernelGGL((rope_tab_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, st, (bf16*)x, ld, rows, T, heads, hd, tab, inverse);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_swiglu_fwd(const void* gu, void* h, long M, int F, int dtype, hipStream_t st) {
    AV_CHECK_ARG(gu && h && M > 0 && F % 4 == 0, "swiglu_fwd: bad args");
    if (dtype == AV_F32) hipLaunchKernelGGL((swiglu_fwd_kernel<float>), dim3(grid_for(M * (F / 4))), dim3(256), 0, st, (const float*)gu, (float*)h, M, F);
    else hipLaunchKernelGGL((swiglu_fwd_kernel<bf16>), dim3(grid_for(M * (F / 4))), dim3(256), 0, st, (const bf16*)gu, (bf16*)h, M, F);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_swiglu_bwd(const void* dh, const void* gu, void* dgu, long M, int F, int dtype, hipStream_t st) {
    AV_CHECK_ARG(dh && gu && dgu && M > 0 && F % 4 == 0, "swiglu_bwd: bad args");
    if (dtype == AV_F32) hipLaunchKernelGGL((swiglu_bwd_kernel<float>), dim3(grid_for(M * (F / 4))), dim3(256), 0, st, (const float*)dh, (const float*)gu, (float*)dgu, M, F);
    else hipLaunchKernelGGL((swiglu_bwd_kernel<bf16>), dim3(grid_for(M * (F / 4))), dim3(256), 0, st, (const bf16*)dh, (const bf16*)gu, (bf16*)dgu, M, F);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_act_residual(const void* x, const void* r, void* y, long n, int act, int dtype, hipStream_t st) {
    AV_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "act_residual: bad args (n=%ld must be a multiple of 4)", n);
    if (dtype == AV_F32) hipLaunchKernelGGL((act_residual_kernel<float>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const float*)x, (const float*)r, (float*)y, n / 4, act);
    else hipLaunchKernelGGL((act_residual_kernel<bf16>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const bf16*)x, (const bf16*)r, (bf16*)y, n / 4, act);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_dropout(const void* x, void* y, long rows, int d, uint32_t seed, float p, int dtype, hipStream_t st, const uint32_t* seed_dev) {
    AV_CHECK_ARG(x && y && rows > 0 && d % 4 == 0 && p >= 0.f && p < 1.f, "dropout: bad args");
    const long total = rows * (d / 4);
    if (dtype == AV_F32) hipLaunchKernelGGL((dropout_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, (float*)y, rows, d, seed, p, seed_dev);
    else hipLaunchKernelGGL((dropout_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, st, (const bf16*)x, (bf16*)y, rows, d, seed, p, seed_dev);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_embedding(const void* table, const int64_t* ids, void* out, long n, int d, int dtype, hipStream_t st) {
    AV_CHECK_ARG(table && ids && out && n > 0 && d % 4 == 0, "embedding: bad args");
    if (dtype == AV_F32) hipLaunchKernelGGL((embedding_kernel<float>), dim3(grid_for(n * (d / 4))), dim3(256), 0, st, (const float*)table, ids, (float*)out, n, d);
    else hipLaunchKernelGGL((embedding_kernel<bf16>), dim3(grid_for(n * (d / 4))), dim3(256), 0, st, (const bf16*)table, ids, (bf16*)out, n, d);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_cast(const void* src, int sdt, void* dst, int ddt, long n, hipStream_t st) {
    AV_CHECK_ARG(src && dst && n > 0, "cast: bad args");
    const dim3 g(grid_for(n / 4 + 1)), b(256);
    if (sdt == AV_F32 && ddt == AV_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), g, b, 0, st, (const float*)src, (bf16*)dst, n);
    else if (sdt == AV_BF16 && ddt == AV_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), g, b, 0, st, (const bf16*)src, (float*)dst, n);
    else if (sdt == AV_F32 && ddt == AV_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, n);
    else if (sdt == AV_BF16 && ddt == AV_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), g, b, 0, st, (const bf16*)src, (bf16*)dst, n);
    else return av_set_error(AV_ERR_ARG, "cast: dtype %d->%d", sdt, ddt);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_whisper_im2col1(const float* mel, void* cols, int B, int n_mels, int T, int Kpad, int dtype, hipStream_t st) {
    AV_CHECK_ARG(mel && cols && B > 0 && T > 0, "im2col1: bad args");
    AV_CHECK_ARG(Kpad % 4 == 0 && Kpad >= n_mels * 3, "im2col1: Kpad=%d too small", Kpad);
    const size_t sh = (size_t)n_mels * 68 * sizeof(float);
    if (dtype == AV_F32) hipLaunchKernelGGL((im2col1_kernel<float>), dim3(av_cdiv(T, 64), B), dim3(256), sh, st, mel, (float*)cols, n_mels, T, Kpad);
    else hipLaunchKernelGGL((im2col1_kernel<bf16>), dim3(av_cdiv(T, 64), B), dim3(256), sh, st, mel, (bf16*)cols, n_mels, T, Kpad);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_whisper_im2col2(const void* h, void* cols, int B, int T, int d, int dtype, hipStream_t st) {
    AV_CHECK_ARG(h && cols && B > 0 && T % 2 == 0 && d % 8 == 0, "im2col2: bad args");
    const long total = (long)B * (T / 2) * 3 * (d / 8);
    if (dtype == AV_F32) hipLaunchKernelGGL((im2col2_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)h, (float*)cols, B, T, d);
    else hipLaunchKernelGGL((im2col2_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, st, (const bf16*)h, (bf16*)cols, B, T, d);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_clip_patchify(const void* frames, void* cols, int N, int S, int p, int Kpad, int dtype, hipStream_t st, int in_dtype) {
    AV_CHECK_ARG(frames && cols && N > 0, "patchify: bad args");
    AV_CHECK_ARG(p > 0 && S % p == 0, "patchify: image %d is not a whole number of %d-pixel patches", S, p);
    const int K = 3 * p * p;
    AV_CHECK_ARG(Kpad >= K && Kpad % 4 == 0, "patchify: Kpad=%d < %d", Kpad, K);
    const int V = p % 4 == 0 ? 4 : p % 2 == 0 ? 2 : 1;
    const long total = (long)N * 3 * S * (S / V);
    const long rows = (long)N * (S / p) * (S / p);
#define AV_PATCHIFY(T, VV) do { if (in_dtype == AV_BF16) hipLaunchKernelGGL((patchify_kernel<T, VV, bf16>), dim3(grid_for(total)), dim3(256), 0, st, (const bf16*)frames, (T*)cols, N, S, p, Kpad); \
                                else hipLaunchKernelGGL((patchify_kernel<T, VV, float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)frames, (T*)cols, N, S, p, Kpad); } while (0)
    if (dtype == AV_F32) {
        if (V == 4) AV_PATCHIFY(float, 4); else if (V == 2) AV_PATCHIFY(float, 2); else AV_PATCHIFY(float, 1);
        if (Kpad > K) hipLaunchKernelGGL((patchify_pad_kernel<float>), dim3(grid_for(rows * (Kpad - K))), dim3(256), 0, st, (float*)cols, rows, K, Kpad);
    } else {
        if (V == 4) AV_PATCHIFY(bf16, 4); else if (V == 2) AV_PATCHIFY(bf16, 2); else AV_PATCHIFY(bf16, 1);
        if (Kpad > K) hipLaunchKernelGGL((patchify_pad_kernel<bf16>), dim3(grid_for(rows * (Kpad - K))), dim3(256), 0, st, (bf16*)cols, rows, K, Kpad);
    }
#undef AV_PATCHIFY
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_clip_cls_rows(const void* class_emb, const void* pos, void* x, int N, int tokens, int d, int dtype, hipStream_t st) {
    AV_CHECK_ARG(class_emb && pos && x && N > 0 && d % 4 == 0, "cls_rows: bad args");
    const long total = (long)N * (d / 4);
    if (dtype == AV_F32) hipLaunchKernelGGL((cls_rows_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)class_emb, (const float*)pos, (float*)x, N, tokens, d);
    else hipLaunchKernelGGL((cls_rows_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, st, (const bf16*)class_emb, (const bf16*)pos, (bf16*)x, N, tokens, d);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_fuse_pool(const void* a, int Ta, const void* v, int Tv, const void* prompt_emb, int P, void* out, int B, int L,
                 int S_out, int D, float fs, int dtype, hipStream_t st) {
    AV_CHECK_ARG((a || v) && out && B > 0 && L > 0 && S_out > 0 && D % 4 == 0, "fuse_pool: bad args");
    AV_CHECK_ARG(P == 0 || prompt_emb, "fuse_pool: P>0 needs prompt_emb");
    const dim3 grid(S_out, B), block(D / 4 < 256 ? (D / 4 + 63) / 64 * 64 : 256);
    if (dtype == AV_F32) hipLaunchKernelGGL((fuse_pool_kernel<float>), grid, block, 0, st, (const float*)a, Ta, (const float*)v, Tv, (const float*)prompt_emb, P, (float*)out, L, S_out, D, fs);
    else hipLaunchKernelGGL((fuse_pool_kernel<bf16>), grid, block, 0, st, (const bf16*)a, Ta, (const bf16*)v, Tv, (const bf16*)prompt_emb, P, (bf16*)out, L, S_out, D, fs);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_lora_pack(const float* A, const float* Bm, int r, int din, int dout, void* A_pad, void* AT_pad, long ld_at,
                 void* B_pad, void* BT_pad, int dtype, hipStream_t st) {
    AV_CHECK_ARG(A && Bm && A_pad && AT_pad && B_pad && BT_pad, "lora_pack: null");
    AV_CHECK_ARG(r > 0 && r <= AVLLM_LORA_PAD, "lora_pack: rank %d > %d unsupported", r, AVLLM_LORA_PAD);
    const long total = (long)AVLLM_LORA_PAD * (din + dout);
    if (dtype == AV_F32) hipLaunchKernelGGL((lora_pack_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, A, Bm, r, din, dout, (float*)A_pad, (float*)AT_pad, ld_at, (float*)B_pad, (float*)BT_pad);
    else hipLaunchKernelGGL((lora_pack_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, st, A, Bm, r, din, dout, (bf16*)A_pad, (bf16*)AT_pad, ld_at, (bf16*)B_pad, (bf16*)BT_pad);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int avllm_lora_pack_batch(const avllm_lora_pack_item* items_dev, int32_t n, int32_t r, int32_t din, int32_t dtype, void* stream) {
    AV_CHECK_ARG(items_dev && n > 0 && din > 0, "lora_pack_batch: null/empty");
    AV_CHECK_ARG(r > 0 && r <= AVLLM_LORA_PAD, "lora_pack_batch: rank %d > %d unsupported", r, AVLLM_LORA_PAD);
    const dim3 grid(n, 16);
    if (dtype == AV_F32) hipLaunchKernelGGL((lora_pack_batch_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, items_dev, r, din);
    else hipLaunchKernelGGL((lora_pack_batch_kernel<bf16>), grid, dim3(256), 0, (hipStream_t)stream, items_dev, r, din);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_kv_append(const void* k, const void* v, long ld, void* kc, void* vc, int B, int T, int pos0, int Tmax, int d,
                 int dtype, hipStream_t st) {
    AV_CHECK_ARG(k && v && kc && vc && pos0 + T <= Tmax && d % 4 == 0, "kv_append: bad args (pos0=%d T=%d Tmax=%d)", pos0, T, Tmax);
    const long total = (long)B * T * (d / 4);
    if (dtype == AV_F32) hipLaunchKernelGGL((kv_append_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)k, (const float*)v, ld, (float*)kc, (float*)vc, B, T, pos0, Tmax, d);
    else hipLaunchKernelGGL((kv_append_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, st, (const bf16*)k, (const bf16*)v, ld, (bf16*)kc, (bf16*)vc, B, T, pos0, Tmax, d);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
