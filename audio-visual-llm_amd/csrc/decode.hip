// Greedy-decode token step (clip_whisper_model.py:1337-1340 -> GenerationMixin greedy search, one new token per sequence on a KV cache):
// every frozen weight is streamed from HBM exactly once per step whatever the batch, so the step is a chain of
// HBM-bound weight streams and its floor is weight bytes / HBM rate.  Round 1 ran 10 launches per decoder layer (norm, q|k|v, RoPE,
// cache append, attention, o, norm, gate|up, SwiGLU, down), five of them on the ~5 us launch floor; here a layer is 5 launches:
//
//   dec_proj<NORM, qkv>      RMSNorm folded into the A operand, q|k|v projection, RoPE on q/k in the accumulators, k/v written
//                            straight into the cache row of this position
//   attn_decode1             one pass over the cache rows (online softmax), up to 16 waves per (sequence, head)
//   dec_proj<plain + R>      o projection + residual
//   dec_proj<NORM, SwiGLU>   RMSNorm folded in, gate|up projection, silu(gate)*up in the epilogue
//   dec_proj<plain + R>      down projection + residual
//
// dec_proj: M <= 16 activation rows ride as one MFMA operand, a workgroup owns 16 weight rows (a contiguous 16*K*2-byte block of
// HBM), its 8 waves split K, each wave keeps a ring of DEPTH 1 KiB weight loads in flight (non-temporal: the stream must not evict
// the activations from L2) and the 8 partial 16x16 tiles meet in LDS.
#include "common.h"
#include "avllm_internal.h"
#include <type_traits>

namespace {

constexpr int DW = 8;            // waves per workgroup (K split)

enum { DEC_PLAIN = 0, DEC_SWIGLU = 1, DEC_QKV = 2 };

struct DecArgs {
    const bf16* A; long lda;            // activations [M, K]
    const bf16* W; long ldw;            // weight rows [*, K]
    const bf16* norm_w; float eps;      // NORM: A is RMS-normalised on the fly (x * rstd * w)
    int M, K, N, mode;
    void* C; long ldc; int out_f32;     // PLAIN: C[M,N] (+R); SWIGLU: C[M,N=F] = silu(gate) * up; QKV: q part [M, dq]
    const bf16* R; long ldr;
    int F;                              // SWIGLU: weight row of "up" column n is F + n
    int dq, dkv, hd;                    // QKV: columns [0,dq) q, [dq,dq+dkv) k, [dq+dkv, dq+2dkv) v; rotary pairs (i, i + hd/2) inside each head
    const float* rope;                  // [hd/2][2] cos,sin of this position
    bf16* kc; bf16* vc;                 // cache of this layer [M(=B), Tmax, dkv]
    int Tmax, pos; const int* pos_dev;  // row written = pos (+ *pos_dev)
    // LoRA side term (peft lora.Linear: + scale * B (A x)): lt [M, ldlt] f32 holds the rank-side products A x of this projection's input
    // (columns 64 j .. 64 j + r of module j; made by a PLAIN launch over the A images), lb[j] the padded B image [rows, 64] of module j
    // (QKV: j = q, k, v; otherwise j = 0).  Added in the epilogue, before RoPE.
    const float* lt; long ldlt; const bf16* lb[3]; float lscale; int lr;
};

// fragment row fr (0..15) of workgroup b -> weight row, and the logical output column it produces
__device__ __forceinline__ int dec_wrow(const DecArgs& a, int b, int fr, int& col) {
    if (a.mode == DEC_SWIGLU) {
        col = 8 * b + (fr & 7);
        return fr < 8 ? col : a.F + col;
    }
    if (a.mode == DEC_QKV) {
        const int c0 = 16 * b;
        if (c0 < a.dq + a.dkv) {        // rotary region: 8 columns of the first half of a head + their 8 partners
            const int per_head = a.hd >> 4;
            const int h = b / per_head, s = b - h * per_head;
            col = h * a.hd + 8 * s + (fr & 7) + (fr < 8 ? 0 : a.hd >> 1);
            return col;
        }
    }
    col = 16 * b + fr;
    return col;
}

// 16-byte global loads the compiler's wait-count pass does not see: the ring below keeps 8 K-steps per wave in flight and waits with
// exact vmcnt values (loads return in issue order), which clang does not do for a register ring (it drains to vmcnt(0) every trip).
typedef u32x4 frag;
template <int OFF> __device__ __forceinline__ void gld(frag& r, const bf16* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void gld_nt(frag& r, const bf16* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm(frag& a, frag& b, frag& c) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}
template <int N> __device__ __forceinline__ void wait_vm(frag& a, frag& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// DPP moves inside each row of 16 lanes (the fr index of an MFMA operand): rotate so that lane i reads lane i + S, or broadcast lane J
template <int CTRL> __device__ __forceinline__ frag dpp4(frag v) {
    frag r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (unsigned)__builtin_amdgcn_update_dpp((int)v[e], (int)v[e], CTRL, 0xf, 0xf, false);
    return r;
}
template <int S> __device__ __forceinline__ frag row_from_higher(frag v) { return dpp4<0x120 + ((16 - S) & 15)>(v); }      // row_ror:(16 - S)
template <int J> __device__ __forceinline__ frag row_bcast(frag v) { return dpp4<0x150 + J>(v); }                           // row_newbcast:J

// One ring group = 4 K-steps of 32.  The weight stream is NOT what limits a projection: a pure read of the same bytes in the same pattern
// runs at 5.6 TB/s, adding one activation load per weight load drops it to 4.0 (tools/ubench/stream_features.hip): the CU's vector-memory
// path is the limiter, so every load that is not a weight load has to go.  AL = activation loads per group:
//   M > 8: 4 (lane (fr, fq) = row fr of one step);  M <= 8: 2 (lanes fr >= 8 fetch rows 0..7 of the NEXT step and a row rotate brings them
//   down when that step is consumed; MFMA output columns >= 8 are garbage nobody reads);  M <= 4: 1 (four steps per load).
// The norm weights of the 4 steps come in ONE load (lane row fr & 3 holds step fr & 3) and reach all rows through a row broadcast.
template <int AL, bool NORM> struct DecGrp { frag wb[4], xa[AL], gw; };

template <bool NORM, int AL, bool LORA>
__global__ __launch_bounds__(DW * 64, 2) void dec_proj_kernel(DecArgs a) {
    __shared__ float part[DW][16][17];      // [wave][n][m]
    __shared__ float ssq[DW][16];
    __shared__ float fin[16][17];           // [m][n]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int M = a.M;
    constexpr int SPL = 4 / AL;              // K-steps per activation load
    constexpr int RPL = 16 / SPL;            // activation rows per load
    const int arow = fr & (RPL - 1), asub = fr / RPL;
    const int ar = arow < M ? arow : M - 1;
    // K in units of 4 steps (128 columns), dealt to the 8 waves as evenly as whole units allow (K = 11008: 11,11,11,11,11,11,10,10)
    const int U = a.K >> 7, ub = U / DW, ue = U - ub * DW;
    const int G = ub + (w < ue ? 1 : 0);
    const long k0 = ((long)w * ub + (w < ue ? w : ue)) << 7;
    int col;
    const int wr = dec_wrow(a, blockIdx.x, fr, col);
    const bf16* ap = a.A + (long)ar * a.lda + k0 + asub * 32 + fq * 8;
    const bf16* bp = a.W + (long)wr * a.ldw + k0 + fq * 8;
    const bf16* gp = NORM ? a.norm_w + k0 + (fr & 3) * 32 + fq * 8 : a.A;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    constexpr int LG = 4 + AL + (NORM ? 1 : 0);          // loads per group, issued as: [norm] then per step: [activation if the step starts a load] weight
    typedef DecGrp<AL, NORM> Grp;
    Grp ga, gb;
    auto issue = [&](Grp& g, int grp) {
        if constexpr (NORM) gld<0>(g.gw, gp + (long)grp * 128);
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j % SPL == 0) gld<64 * j>(g.xa[j / SPL], ap + (long)grp * 128);
            gld_nt<64 * j>(g.wb[j], bp + (long)grp * 128);
        });
    };
    auto step = [&](Grp& g, auto jc, auto behind) {      // wait for step j of this group (`behind` younger groups in flight), multiply
        constexpr int j = decltype(jc)::value;
        constexpr int through = (NORM ? 1 : 0) + (j / SPL + 1) + (j + 1);            // loads of this group issued up to and including weight j
        constexpr int N = LG - through + decltype(behind)::value * LG;
        if constexpr (NORM) wait_vm<N>(g.wb[j], g.xa[j / SPL], g.gw);
        else wait_vm<N>(g.wb[j], g.xa[j / SPL]);
        frag xr = g.xa[j / SPL];
        if constexpr (j % SPL != 0) xr = row_from_higher<RPL * (j % SPL)>(xr);
        bf16x8 x = __builtin_bit_cast(bf16x8, xr);
        if constexpr (NORM) {
            const bf16x8 gv = __builtin_bit_cast(bf16x8, row_bcast<j>(g.gw));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)x[e];
                ss += xf * xf;
                x[e] = (bf16)(xf * (float)gv[e]);
            }
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, g.wb[j]), x, acc, 0, 0, 0);      // D[n][m]
    };
    auto consume = [&](Grp& g, auto behind) { static_for<0, 4>([&](auto jc) { step(g, jc, behind); }); };
    const std::integral_constant<int, 0> none{};
    const std::integral_constant<int, 1> one{};
    // Two groups in flight; a group is refilled in one burst once its 4 steps are consumed.  Single-exit loops with fixed slot roles (a loop
    // whose exit alternates between the groups makes the compiler copy ring registers that still have loads in flight): an odd group count
    // peels one refill first, which swaps the roles for the rest of the wave's life.
    if (G == 1) {
        issue(ga, 0);
        consume(ga, none);
    } else if (G >= 2 && !(G & 1)) {
        issue(ga, 0);
        issue(gb, 1);
        for (int g = 0; g + 2 < G; g += 2) { consume(ga, one); issue(ga, g + 2); consume(gb, one); issue(gb, g + 3); }
        consume(ga, one);
        consume(gb, none);
    } else if (G >= 3) {
        issue(ga, 0);
        issue(gb, 1);
        consume(ga, one); issue(ga, 2);
        for (int g = 1; g + 2 < G; g += 2) { consume(gb, one); issue(gb, g + 2); consume(ga, one); issue(ga, g + 3); }
        consume(gb, one);
        consume(ga, none);
    }
    // adapters: the finishing thread (m, nn) fetches its 16 rank-side products and its B row now (the weight ring has drained; the loads
    // fly while the partial tiles meet in LDS)
    f32x4 ltv[4] = {};
    bf16x8 lbv[2] = {};
    if constexpr (LORA) {
        if (threadIdx.x < 256 && (int)(threadIdx.x >> 4) < M) {
            int oc;
            dec_wrow(a, blockIdx.x, threadIdx.x & 15, oc);
            int j = 0, row = oc;
            if (a.mode == DEC_QKV) { j = oc < a.dq ? 0 : (oc < a.dq + a.dkv ? 1 : 2); row = oc - (j == 0 ? 0 : (j == 1 ? a.dq : a.dq + a.dkv)); }
            if (oc < a.N) {
                const float* tp = a.lt + (long)(threadIdx.x >> 4) * a.ldlt + 64 * j;
                const bf16* bp2 = a.lb[j] + (long)row * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i) ltv[i] = *(const f32x4*)(tp + 4 * i);
                lbv[0] = *(const bf16x8*)bp2;
                lbv[1] = *(const bf16x8*)(bp2 + 8);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) part[w][fq * 4 + i][fr] = acc[i];
    if (NORM) {
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (fq == 0) ssq[w][fr] = ss;
    }
    __syncthreads();
    if (threadIdx.x >= 256) return;
    const int m = threadIdx.x >> 4, nn = threadIdx.x & 15;
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < DW; ++x) s += part[x][nn][m];
    if (NORM) {
        float t = 0.f;
#pragma unroll
        for (int x = 0; x < DW; ++x) t += ssq[x][m];
        s *= rsqrtf(t / (float)a.K + a.eps);
    }
    if constexpr (LORA) {
        float u = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) u += (i < a.lr ? ltv[i >> 2][i & 3] : 0.f) * (float)lbv[i >> 3][i & 7];
        s += a.lscale * u;
    }
    int ocol;
    dec_wrow(a, blockIdx.x, nn, ocol);
    if (a.mode == DEC_PLAIN) {
        if (m < M && ocol < a.N) {
            if (a.R) s += (float)a.R[(long)m * a.ldr + ocol];
            if (a.out_f32) ((float*)a.C)[(long)m * a.ldc + ocol] = s;
            else ((bf16*)a.C)[(long)m * a.ldc + ocol] = (bf16)s;
        }
        return;
    }
    fin[m][nn] = s;
    __syncthreads();                         // all 256 remaining threads reach it (4 whole waves)
    if (m >= M) return;
    const float other = fin[m][nn ^ 8];
    if (a.mode == DEC_SWIGLU) {
        if (nn < 8) ((bf16*)a.C)[(long)m * a.ldc + ocol] = (bf16)(s / (1.0f + __expf(-s)) * other);
        return;
    }
    // DEC_QKV
    const int pos = a.pos + (a.pos_dev ? *a.pos_dev : 0);
    float o = s;
    if (ocol < a.dq + a.dkv) {
        const int i = (ocol % a.hd) & ((a.hd >> 1) - 1);
        const float c = a.rope[2 * i], sn = a.rope[2 * i + 1];
        o = nn < 8 ? s * c - other * sn : s * c + other * sn;
    }
    // position from device memory: the host could not check it.  A step replayed past the end of the cache writes nothing (the cache
    // rows of the last valid position stay as they are) instead of running over [B, Tmax, dkv]; attn_decode1_kernel clamps its length alike.
    const bool in_cache = pos >= 0 && pos < a.Tmax;
    if (ocol < a.dq) ((bf16*)a.C)[(long)m * a.ldc + ocol] = (bf16)o;
    else if (!in_cache) return;
    else if (ocol < a.dq + a.dkv) a.kc[((long)m * a.Tmax + pos) * a.dkv + ocol - a.dq] = (bf16)o;
    else a.vc[((long)m * a.Tmax + pos) * a.dkv + ocol - a.dq - a.dkv] = (bf16)o;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Single-token attention over the cache, ONE pass: a group of G = hd/8 lanes owns cache rows t = g, g + R, ... and carries a running
// (max, sum, weighted V) triple; K and V rows of a trip are all requested before the first is used.  Groups are merged through
// shuffles (same wave) and LDS (NW waves).  No score buffer: Tk is unbounded and may come from device memory.
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode1_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ kc, const T* __restrict__ vc,
                                                               T* __restrict__ o, long ldo, int H, int hd, int Tk, const int* __restrict__ tk_dev,
                                                               int Tmax, float scale, int GQ) {
    __shared__ float pacc[NW][128];
    __shared__ float pm[NW], pl[NW];
    const int h = blockIdx.x, b = blockIdx.y, d = (H / GQ) * hd;
    if (tk_dev) Tk += *tk_dev;
    Tk = Tk > Tmax ? Tmax : Tk;                      // a device-side length the host could not check never reads past the cache
    const int G = hd >> 3, R = NW * 64 / G;
    const int tid = threadIdx.x, dc = tid % G, rsub = tid / G;
    float qv[8];
    load_f<8>(q + (long)b * ldq + (long)h * hd + dc * 8, qv);
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] *= scale;
    const T* kbase = kc + ((long)b * Tmax) * d + (long)(h / GQ) * hd + dc * 8;
    const T* vbase = vc + ((long)b * Tmax) * d + (long)(h / GQ) * hd + dc * 8;
    constexpr int UN = 4;
    float mx = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < Tk; t0 += UN * R) {
        float kv[UN][8], vv[UN][8];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int t = t0 + u * R + rsub;
#pragma unroll
            for (int j = 0; j < 8; ++j) { kv[u][j] = 0.f; vv[u][j] = 0.f; }
            if (t < Tk) { load_f<8>(kbase + (long)t * d, kv[u]); load_f<8>(vbase + (long)t * d, vv[u]); }
        }
        float s[UN], tm = mx;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int t = t0 + u * R + rsub;
            float x = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) x += qv[j] * kv[u][j];
            for (int off = G >> 1; off > 0; off >>= 1) x += __shfl_xor(x, off);
            s[u] = t < Tk ? x : -INFINITY;
            tm = fmaxf(tm, s[u]);
        }
        if (tm > -INFINITY) {
            const float f = __expf(mx - tm);             // mx = -inf on the first trip: exp(-inf) = 0, acc and l are 0 anyway
            l *= f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] *= f;
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const float p = __expf(s[u] - tm);       // -inf rows: 0
                l += p;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += p * vv[u][j];
            }
            mx = tm;
        }
    }
    // merge the row groups of a wave (lanes with equal dc)
    for (int off = G; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(mx, off), l2 = __shfl_xor(l, off);
        const float mn = fmaxf(mx, m2);
        const float f1 = mx > -INFINITY ? __expf(mx - mn) : 0.f, f2 = m2 > -INFINITY ? __expf(m2 - mn) : 0.f;
        l = l * f1 + l2 * f2;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = acc[j] * f1 + __shfl_xor(acc[j], off) * f2;
        mx = mn;
    }
    const int w = tid >> 6, lane = tid & 63;
    if (lane < G) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pacc[w][lane * 8 + j] = acc[j];
        if (lane == 0) { pm[w] = mx; pl[w] = l; }
    }
    __syncthreads();
    if (tid < hd) {
        float mn = -INFINITY;
#pragma unroll
        for (int x = 0; x < NW; ++x) mn = fmaxf(mn, pm[x]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int x = 0; x < NW; ++x) {
            const float f = pm[x] > -INFINITY ? __expf(pm[x] - mn) : 0.f;
            num += pacc[x][tid] * f;
            den += pl[x] * f;
        }
        o[(long)b * ldo + (long)h * hd + tid] = from_f<T>(num / den);
    }
}

}  // namespace

bool av_dec_proj_supported(int dtype, int M, int K, int N, int mode, int hd) {
    if (dtype != AV_BF16 || M < 1 || M > 16 || K % 128 != 0) return false;
    if (mode == DEC_SWIGLU) return N % 8 == 0;
    if (mode == DEC_QKV) return N % 16 == 0 && hd % 16 == 0 && hd >= 32;
    return N % 16 == 0;
}

static int dec_launch(DecArgs& a, hipStream_t st) {
    const int grid = a.mode == DEC_SWIGLU ? a.N / 8 : a.N / 16;
#define DEC_LAUNCH(NORMV, ALV)                                                                                                   \
    do {                                                                                                                       \
        if (a.lt) hipLaunchKernelGGL((dec_proj_kernel<NORMV, ALV, true>), dim3(grid), dim3(DW * 64), 0, st, a);               \
        else hipLaunchKernelGGL((dec_proj_kernel<NORMV, ALV, false>), dim3(grid), dim3(DW * 64), 0, st, a);                    \
    } while (0)
    const int ev = av_knob(AV_KNOB_DEC_AL);       // experiment knob: force the activation-load form (4 = one load per step)
    const int al = ev ? ev : (a.M <= 4 ? 1 : a.M <= 8 ? 2 : 4);
    if (a.norm_w) { if (al == 1 && a.M <= 4) DEC_LAUNCH(true, 1); else if (al <= 2 && a.M <= 8) DEC_LAUNCH(true, 2); else DEC_LAUNCH(true, 4); }
    else { if (al == 1 && a.M <= 4) DEC_LAUNCH(false, 1); else if (al <= 2 && a.M <= 8) DEC_LAUNCH(false, 2); else DEC_LAUNCH(false, 4); }
#undef DEC_LAUNCH
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_dec_proj(const avllm_dec_proj_desc* d, hipStream_t st) {
    AV_CHECK_ARG(d && d->A && d->W, "dec_proj: null operand");
    AV_CHECK_ARG(d->mode >= DEC_PLAIN && d->mode <= DEC_QKV, "dec_proj: mode %d", d->mode);
    AV_CHECK_ARG(av_dec_proj_supported(AV_BF16, d->M, d->K, d->N, d->mode, d->hd),
                 "dec_proj: bf16, 1 <= M <= 16 (M=%d), K %% 128 == 0 (K=%d), N %% 16 == 0 (%% 8 for SwiGLU; N=%d)", d->M, d->K, d->N);
    AV_CHECK_ARG(d->lda % 8 == 0 && d->ldw % 8 == 0 && d->lda >= d->K && d->ldw >= d->K, "dec_proj: rows must be 16-byte aligned and hold K elements");
    DecArgs a = {};
    a.A = (const bf16*)d->A; a.lda = d->lda; a.W = (const bf16*)d->W; a.ldw = d->ldw; a.norm_w = (const bf16*)d->norm_w; a.eps = d->eps;
    a.M = d->M; a.K = d->K; a.N = d->N; a.mode = d->mode;
    a.C = d->C; a.ldc = d->ldc; a.out_f32 = d->out_f32; a.R = (const bf16*)d->R; a.ldr = d->ldr;
    if (d->mode == DEC_PLAIN) {
        AV_CHECK_ARG(d->C && d->ldc >= d->N && (!d->R || d->ldr >= d->N), "dec_proj: output [M,N] missing or rows shorter than N");
    } else if (d->mode == DEC_SWIGLU) {
        AV_CHECK_ARG(d->C && d->ldc >= d->N && !d->R && !d->out_f32, "dec_proj(SwiGLU): bf16 output [M,F] without residual");
        a.F = d->N;
    } else {
        AV_CHECK_ARG(d->C && d->kc && d->vc && d->rope && d->dq > 0 && d->dkv > 0 && d->N == d->dq + 2 * d->dkv && d->dq % d->hd == 0 && d->dkv % d->hd == 0 &&
                     d->ldc >= d->dq && !d->R && !d->out_f32, "dec_proj(q|k|v): N=%d must be dq + 2 dkv (dq=%d dkv=%d) in whole heads of %d", d->N, d->dq, d->dkv, d->hd);
        AV_CHECK_ARG(d->Tmax > 0 && d->pos >= 0 && (d->pos_dev || d->pos < d->Tmax), "dec_proj(q|k|v): pos=%d outside the cache (Tmax=%d)", d->pos, d->Tmax);
        a.dq = d->dq; a.dkv = d->dkv; a.hd = d->hd; a.rope = d->rope; a.kc = (bf16*)d->kc; a.vc = (bf16*)d->vc; a.Tmax = d->Tmax; a.pos = d->pos;
        a.pos_dev = d->pos_dev;
    }
    if (d->lora_t) {
        const int nmod = d->mode == DEC_QKV ? 3 : 1;
        AV_CHECK_ARG(d->mode != DEC_SWIGLU && d->lora_r > 0 && d->lora_r <= 16 && d->ld_lora_t >= 64 * nmod && d->ld_lora_t % 4 == 0 &&
                     ((uintptr_t)d->lora_t & 15) == 0, "dec_proj: adapters need rank <= 16 (r=%d), 16-byte aligned rank-side products with >= %d columns per row",
                     d->lora_r, 64 * nmod);
        for (int j = 0; j < nmod; ++j) AV_CHECK_ARG(d->lora_b[j], "dec_proj: adapter B image %d missing", j);
        a.lt = d->lora_t; a.ldlt = d->ld_lora_t; a.lscale = d->lora_scale; a.lr = d->lora_r;
        for (int j = 0; j < nmod; ++j) a.lb[j] = (const bf16*)d->lora_b[j];
    }
    return dec_launch(a, st);
}

extern "C" int avllm_dec_proj(const avllm_dec_proj_desc* d, void* stream) { return av_dec_proj(d, (hipStream_t)stream); }

int av_attention_decode1(const void* q, long ldq, const void* kc, const void* vc, void* o, long ldo, int B, int H, int hd, int Tk, const int* tk_dev,
                         int Tmax, float scale, int dtype, hipStream_t st, int G) {
    AV_CHECK_ARG(q && kc && vc && o && (tk_dev || (Tk > 0 && Tk <= Tmax)) && G > 0 && H % G == 0, "attention_decode: bad args");
    AV_CHECK_ARG((hd == 64 || hd == 128) && ldq % 8 == 0, "attention_decode: head_dim %d unsupported", hd);
    const dim3 grid(H, B);
    // few (sequence, head) pairs: 16 waves each so that a CU still has ~64 KiB of cache rows in flight
    if ((long)B * H <= 1024) {
        if (dtype == AV_F32) hipLaunchKernelGGL((attn_decode1_kernel<float, 16>), grid, dim3(1024), 0, st, (const float*)q, ldq, (const float*)kc, (const float*)vc, (float*)o, ldo, H, hd, Tk, tk_dev, Tmax, scale, G);
        else hipLaunchKernelGGL((attn_decode1_kernel<bf16, 16>), grid, dim3(1024), 0, st, (const bf16*)q, ldq, (const bf16*)kc, (const bf16*)vc, (bf16*)o, ldo, H, hd, Tk, tk_dev, Tmax, scale, G);
    } else {
        if (dtype == AV_F32) hipLaunchKernelGGL((attn_decode1_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)q, ldq, (const float*)kc, (const float*)vc, (float*)o, ldo, H, hd, Tk, tk_dev, Tmax, scale, G);
        else hipLaunchKernelGGL((attn_decode1_kernel<bf16, 4>), grid, dim3(256), 0, st, (const bf16*)q, ldq, (const bf16*)kc, (const bf16*)vc, (bf16*)o, ldo, H, hd, Tk, tk_dev, Tmax, scale, G);
    }
    AV_LAUNCH_CHECK();
    return AV_OK;
}

__global__ void pos_advance_kernel(int* p, int by) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += by; }

extern "C" int avllm_pos_advance(int32_t* pos_dev, int32_t by, void* stream) {
    AV_CHECK_ARG(pos_dev, "pos_advance: null");
    hipLaunchKernelGGL(pos_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pos_dev, by);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int avllm_attention_decode(const void* q, int64_t ldq, const void* kc, const void* vc, void* o, int64_t ldo, int32_t B, int32_t H, int32_t hd,
                                      int32_t Tk, const int32_t* tk_dev, int32_t Tmax, float scale, int32_t kv_group, int32_t dtype, void* stream) {
    AV_CHECK_ARG(B > 0 && H > 0 && (dtype == AV_F32 || dtype == AV_BF16), "attention_decode: B=%d H=%d dtype=%d", B, H, dtype);
    return av_attention_decode1(q, ldq, kc, vc, o, ldo, B, H, hd, Tk, tk_dev, Tmax, scale, dtype, (hipStream_t)stream, kv_group);
}
