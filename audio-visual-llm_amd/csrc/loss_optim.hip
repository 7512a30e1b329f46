// Shifted causal-LM cross entropy (fwd statistics + in-place gradient), row argmax (greedy decode),
// clip_grad_norm_ + AdamW on the flat LoRA buffer, and the LoRA dA/dB reduction GEMM.
// Reference: HF:loss/loss_utils.py:49-71 (ForCausalLMLoss), trainer/clip_whisper_trainer.py:457-464,171-232.
#include "common.h"
#include "avllm_internal.h"

int av_gemm_tn_mfma(const void* Big, long ldb, int NB, const void* Small, long lds_, int R, int M, float* out, long ldo, float alpha,
                    int trans_out, hipStream_t st, uint32_t drop_seed, float drop_p, const uint32_t* seed_dev);

namespace {

template <typename T> struct VecN { static constexpr int n = 4; };
template <> struct VecN<bf16> { static constexpr int n = 8; };

// one 256-thread block per row: online (max, sum exp) over V in 16-byte vectors
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, long ld, const int64_t* __restrict__ labels,
                                                     int Tn, int V, float* __restrict__ row_lse, float* __restrict__ loss_sum,
                                                     float* __restrict__ count) {
    __shared__ float red[8];
    constexpr int VN = VecN<T>::n;
    const long row = blockIdx.x;
    const int t = (int)(row % Tn);
    const long b = row / Tn;
    const int64_t tgt = (t + 1 < Tn) ? labels[b * Tn + t + 1] : -100;
    const T* lr = logits + row * ld;
    float mx = -INFINITY, sm = 0.f;
    for (int c = threadIdx.x * VN; c < V; c += 256 * VN) {
        float v[VN];
        if (c + VN <= V) load_f<VN>(lr + c, v);
        else { for (int j = 0; j < VN; ++j) v[j] = c + j < V ? to_f(lr[c + j]) : -INFINITY; }
        float lm = v[0];
#pragma unroll
        for (int j = 1; j < VN; ++j) lm = fmaxf(lm, v[j]);
        const float nm = fmaxf(mx, lm);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VN; ++j) s += __expf(v[j] - nm);
        sm = sm * __expf(mx - nm) + s;
        mx = nm;
    }
    const float gmx = block_max(mx, red);
    const float gs = block_sum(sm * __expf(mx - gmx), red);
    if (threadIdx.x == 0) {
        const float lse = gmx + logf(gs);
        row_lse[row] = lse;
        if (tgt >= 0 && tgt < V) {
            atomicAdd(loss_sum, lse - to_f(lr[tgt]));
            atomicAdd(count, 1.0f);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, long ld, const int64_t* __restrict__ labels,
                                                     const float* __restrict__ row_lse, const float* __restrict__ count,
                                                     float grad_scale, T* __restrict__ dl, int Tn, int V) {
    constexpr int VN = VecN<T>::n;
    const long row = blockIdx.x;
    const int t = (int)(row % Tn);
    const long b = row / Tn;
    const int64_t tgt = (t + 1 < Tn) ? labels[b * Tn + t + 1] : -100;
    const bool scored = tgt >= 0 && tgt < V;
    const float cnt = count[0];
    const float g = scored && cnt > 0.f ? grad_scale / cnt : 0.f;
    const float lse = row_lse[row];
    const T* lr = logits + row * ld;
    T* dr = dl + row * ld;
    for (int c = threadIdx.x * VN; c < V; c += 256 * VN) {
        float v[VN];
        if (c + VN <= V) {
            load_f<VN>(lr + c, v);
#pragma unroll
            for (int j = 0; j < VN; ++j) v[j] = scored ? (__expf(v[j] - lse) - ((int64_t)(c + j) == tgt ? 1.f : 0.f)) * g : 0.f;
            store_f<VN>(dr + c, v);
        } else {
            for (int j = 0; j < VN && c + j < V; ++j) {
                const float x = to_f(lr[c + j]);
                dr[c + j] = from_f<T>(scored ? (__expf(x - lse) - ((int64_t)(c + j) == tgt ? 1.f : 0.f)) * g : 0.f);
            }
        }
    }
}

// first index of the maximum (torch.argmax tie rule on exact ties: lowest index)
template <typename T>
__global__ __launch_bounds__(256) void argmax_kernel(const T* __restrict__ x, long ld, int V, int64_t* __restrict__ out) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const T* r = x + (long)blockIdx.x * ld;
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int c = threadIdx.x; c < V; c += 256) { const float v = to_f(r[c]); if (v > best) { best = v; idx = c; } }
    bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float ov = bv[threadIdx.x + s]; const int oi = bi[threadIdx.x + s];
            if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ov; bi[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = bi[0];
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
    __shared__ float red[8];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(g + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// The same sum in a FIXED order: block b writes its partial to partials[b] (block-internal order is fixed by block_sum), then one block adds
// the partials in index order.  Data-parallel replicas compute the clip coefficient from identical all-reduced gradients: with float atomics
// (sumsq_kernel) the last bits of the norm depend on arrival order, the replicas' parameters drift apart by ulps per step; with this form
// they stay bit-identical.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partials) {
    __shared__ float red[8];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(g + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partials, int nb, float* __restrict__ out) {
    __shared__ float red[8];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partials[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

// torch.optim.AdamW single-tensor rule (decoupled decay first), preceded by clip_grad_norm_'s scaling
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                             const float* __restrict__ sumsq, float max_norm, float prescale, const float* __restrict__ guard,
                             float* __restrict__ skipped, avllm_step_state* __restrict__ state) {
    if (state) { lr = state->lr; bc1 = state->bc1; bc2_sqrt = state->bc2_sqrt; }     // graph-replayable step: per-step scalars live on the device
    // Non-finite step guard (trainer/clip_whisper_trainer.py:444-452 skips backward and the optimizer on a NaN/Inf loss): decided on the
    // device from values every thread reads alike, so the step costs no host sync and p, m, v stay untouched when it is skipped.
    const bool bad = (sumsq && !isfinite(sumsq[0])) || (guard && !isfinite(guard[0]));
    if (bad) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (skipped) atomicAdd(skipped, 1.0f);
            // a skipped step is not an optimizer step: the reference runs neither optimizer.step() nor scheduler.step() on it
            // (trainer/clip_whisper_trainer.py:444-452), so the step count that drives lr and the bias corrections goes back by one.
            // Nobody else reads state->step in this launch (lr / bc1 / bc2_sqrt were copied above; every other thread returns).
            if (state) { atomicAdd(&state->skipped, 1.0f); if (state->step > 0) state->step -= 1; }
        }
        return;
    }
    float coef = prescale;
    if (sumsq && max_norm > 0.f) {
        const float norm = sqrtf(sumsq[0]) * prescale;
        coef *= fminf(1.0f, max_norm / (norm + 1e-6f));
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = m[i] * b1 + gi * (1.0f - b1);
        const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// out[i,j] += alpha * sum_m P[m,i] Q[m,j].  One of I,J is the LoRA rank (<=64), the other the model width.
// Tile: 64 (i) x 64 (j) outputs per block, M split over gridDim.z chunks, fp32 atomics into `out`
// (<= 32 adders per element; float atomics keep the sum in fp32 -- MI355X_MICROARCH "Global float atomics").
template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const T* __restrict__ P, long ldp, int I, const T* __restrict__ Q, long ldq,
                                                      int J, int M, int mchunk, float* __restrict__ out, long ldo, float alpha) {
    __shared__ float Ps[32][65], Qs[32][65];
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int m_begin = blockIdx.z * mchunk, m_end = min(M, m_begin + mchunk);
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;      // 16x16 threads, 4x4 outputs each
    float acc[4][4] = {};
    for (int mb = m_begin; mb < m_end; mb += 32) {
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * 64; e += 256) {
            const int mm = e >> 6, c = e & 63;
            const int m = mb + mm;
            Ps[mm][c] = (m < m_end && i0 + c < I) ? to_f(P[(long)m * ldp + i0 + c]) : 0.f;
            Qs[mm][c] = (m < m_end && j0 + c < J) ? to_f(Q[(long)m * ldq + j0 + c]) : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int mm = 0; mm < 32; ++mm) {
            float a[4], b[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) { a[x] = Ps[mm][ti * 4 + x]; b[x] = Qs[mm][tj * 4 + x]; }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] += a[x] * b[y];
        }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int i = i0 + ti * 4 + x, j = j0 + tj * 4 + y;
            if (i < I && j < J) atomicAdd(out + (long)i * ldo + j, alpha * acc[x][y]);
        }
}

}  // namespace

int av_ce_fwd(const void* logits, long ld, const int64_t* labels, int B, int T, int V, float* row_lse, float* loss_sum,
              float* count, int dtype, hipStream_t st) {
    AV_CHECK_ARG(logits && labels && row_lse && loss_sum && count && B > 0 && T > 0 && V > 0, "ce_fwd: bad args");
    AV_CHECK_ARG(ld % 8 == 0, "ce_fwd: ld must be a multiple of 8");
    if (dtype == AV_F32) hipLaunchKernelGGL((ce_fwd_kernel<float>), dim3((long)B * T), dim3(256), 0, st, (const float*)logits, ld, labels, T, V, row_lse, loss_sum, count);
    else hipLaunchKernelGGL((ce_fwd_kernel<bf16>), dim3((long)B * T), dim3(256), 0, st, (const bf16*)logits, ld, labels, T, V, row_lse, loss_sum, count);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_ce_bwd(const void* logits, long ld, const int64_t* labels, const float* row_lse, const float* count,
              float grad_scale, void* dlogits, int B, int T, int V, int dtype, hipStream_t st) {
    AV_CHECK_ARG(logits && labels && row_lse && count && dlogits, "ce_bwd: bad args");
    if (dtype == AV_F32) hipLaunchKernelGGL((ce_bwd_kernel<float>), dim3((long)B * T), dim3(256), 0, st, (const float*)logits, ld, labels, row_lse, count, grad_scale, (float*)dlogits, T, V);
    else hipLaunchKernelGGL((ce_bwd_kernel<bf16>), dim3((long)B * T), dim3(256), 0, st, (const bf16*)logits, ld, labels, row_lse, count, grad_scale, (bf16*)dlogits, T, V);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_argmax_rows(const void* logits, long ld, long rows, int V, int64_t* out, int dtype, hipStream_t st) {
    AV_CHECK_ARG(logits && out && rows > 0 && V > 0, "argmax: bad args");
    if (dtype == AV_F32) hipLaunchKernelGGL((argmax_kernel<float>), dim3(rows), dim3(256), 0, st, (const float*)logits, ld, V, out);
    else hipLaunchKernelGGL((argmax_kernel<bf16>), dim3(rows), dim3(256), 0, st, (const bf16*)logits, ld, V, out);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_grad_sumsq(const float* g, long n, float* sumsq, hipStream_t st) {
    AV_CHECK_ARG(g && sumsq && n > 0, "grad_sumsq: bad args");
    long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, st, g, n, sumsq);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_grad_sumsq_det(const float* g, long n, float* partials, int nparts, float* sumsq, hipStream_t st) {
    AV_CHECK_ARG(g && sumsq && partials && n > 0 && nparts >= 1, "grad_sumsq_det: bad args");
    long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > nparts ? nparts : blocks);
    blocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, g, n, partials);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, partials, (int)blocks, sumsq);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                  float wd, int step, const float* sumsq, float max_norm, float grad_prescale, const float* guard, float* skipped,
                  const avllm_step_state* state, hipStream_t st) {
    AV_CHECK_ARG(p && g && m && v && n > 0 && (step >= 1 || state), "adamw: bad args");
    if (step < 1) step = 1;
    const float bc1 = 1.0f - (float)pow((double)b1, step);
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, step));
    long blocks = (n + 255) / 256;
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s, sumsq, max_norm, grad_prescale, guard, skipped, (avllm_step_state*)state);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

// One thread advances the step's device-side scalars (include/avllm.h avllm_step_state): the host-side bookkeeping of
// trainer/clip_whisper_trainer.py:461-464 (optimizer step count, scheduler.step()) and the step's dropout seed.
__global__ void step_advance_kernel(avllm_step_state* s, avllm_schedule c) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t step = s->step + 1;
    const float t = (float)(step - 1), total = (float)(c.total_steps > 0 ? c.total_steps : 1);
    float lr;
    if (c.warmup_steps > 0) {
        const float w = (float)c.warmup_steps;
        if (t < w) lr = c.base_lr * t / w;
        else lr = c.base_lr * fmaxf(0.f, 0.5f * (1.0f + cosf(3.14159265358979323846f * (t - w) / fmaxf(1.f, total - w))));
    } else {
        lr = c.base_lr * (1.0f + cosf(3.14159265358979323846f * t / total)) * 0.5f;
    }
    s->step = step;
    s->lr = lr;
    s->bc1 = 1.0f - powf(c.beta1, (float)step);
    s->bc2_sqrt = sqrtf(1.0f - powf(c.beta2, (float)step));
    // + the skipped count: a step that is retried after a skipped one (same step number) draws fresh masks
    s->dropout_seed = (step + (uint32_t)s->skipped) * 0x9E3779B1u + c.rank * 0x85EBCA6Bu + 12345u;
}

int av_step_advance(avllm_step_state* state, const avllm_schedule* sched, hipStream_t st) {
    AV_CHECK_ARG(state && sched, "step_advance: null");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, st, state, *sched);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_gemm_tn(const void* P, long ldp, int I, const void* Q, long ldq, int J, int M, float* out, long ldo,
               float alpha, int dtype, hipStream_t st, uint32_t drop_seed, float drop_p, const uint32_t* seed_dev) {
    AV_CHECK_ARG(P && Q && out && I > 0 && J > 0 && M > 0, "gemm_tn: bad args");
    if (dtype == AV_BF16 && ldp % 8 == 0 && ldq % 8 == 0) {
        if (J <= 16 && I % 128 == 0) return av_gemm_tn_mfma(P, ldp, I, Q, ldq, J, M, out, ldo, alpha, 0, st, drop_seed, drop_p, seed_dev);
        if (I <= 16 && J % 128 == 0) return av_gemm_tn_mfma(Q, ldq, J, P, ldp, I, M, out, ldo, alpha, 1, st, drop_seed, drop_p, seed_dev);
    }
    if (drop_p > 0.f) return av_set_error(AV_ERR_UNSUPPORTED, "gemm_tn: on-the-fly dropout needs the bf16 MFMA path (rank <= 16, width %% 128 == 0)");
    int zs = av_cdiv(M, 256);
    zs = zs > 32 ? 32 : zs;
    int mchunk = av_cdiv(M, zs);
    mchunk = (mchunk + 31) / 32 * 32;
    zs = av_cdiv(M, mchunk);
    const dim3 grid(av_cdiv(I, 64), av_cdiv(J, 64), zs);
    if (dtype == AV_F32) hipLaunchKernelGGL((gemm_tn_kernel<float>), grid, dim3(256), 0, st, (const float*)P, ldp, I, (const float*)Q, ldq, J, M, mchunk, out, ldo, alpha);
    else hipLaunchKernelGGL((gemm_tn_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)P, ldp, I, (const bf16*)Q, ldq, J, M, mchunk, out, ldo, alpha);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
