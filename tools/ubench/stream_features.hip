// What in a decode projection costs the 35 % between a pure weight stream (5.6 TB/s, stream_pattern.hip) and avllm_dec_proj (3.6 TB/s)?
// The weight stream of pattern 0 with features added one at a time:
//   bit 0: activation loads (16 bytes per lane per K-step from a [8, K] matrix: L2 hits, same instruction count as the weight loads)
//   bit 1: MFMA accumulate (needs bit 0)        bit 2: LDS reduction of the 8 waves + barrier + bf16 store (needs bit 1)
//   bit 3: RMSNorm weight loads + the VALU work of folding the norm into the operand (needs bit 1)
//   bit 4: residual load at the end (needs bit 2)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int F>
__global__ __launch_bounds__(512) void feat_kernel(const __bf16* __restrict__ W, const __bf16* __restrict__ A, const __bf16* __restrict__ G, int K, int N,
                                                   __bf16* __restrict__ out, const __bf16* __restrict__ R) {
    __shared__ float part[8][16][17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const long n0 = (long)blockIdx.x * 16;
    const int kw = K / 8, nk = kw / 32;
    const __bf16* bp = W + (n0 + fr) * K + (long)w * kw + fq * 8;
    const __bf16* ap = A + (long)(fr & 7) * K + (long)w * kw + fq * 8;
    const __bf16* gp = G + (long)w * kw + fq * 8;
    u32x4 x = {0, 0, 0, 0};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    for (int s = 0; s < nk; s += 8) {
        u32x4 v[8], a[8], g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = s + u < nk;
            v[u] = ok ? *(const u32x4*)(bp + (s + u) * 32) : (u32x4){0, 0, 0, 0};
            if (F & 1) a[u] = ok ? *(const u32x4*)(ap + (s + u) * 32) : (u32x4){0, 0, 0, 0};
            if (F & 8) g[u] = ok ? *(const u32x4*)(gp + (s + u) * 32) : (u32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (F & 2) {
                bf16x8 xa = __builtin_bit_cast(bf16x8, a[u]);
                if (F & 8) {
                    const bf16x8 gv = __builtin_bit_cast(bf16x8, g[u]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float xf = (float)xa[e]; ss += xf * xf; xa[e] = (__bf16)(xf * (float)gv[e]); }
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v[u]), xa, acc, 0, 0, 0);
            } else {
                x ^= v[u];
                if (F & 1) x ^= a[u];
            }
        }
    }
    if (F & 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) part[w][fq * 4 + i][fr] = acc[i] + ss;
        __syncthreads();
        if (threadIdx.x < 256) {
            const int m = threadIdx.x >> 4, nn = threadIdx.x & 15;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += part[q][nn][m];
            if (m < 8) {
                if (F & 16) s += (float)R[(long)m * N + n0 + nn];
                out[(long)m * N + n0 + nn] = (__bf16)s;
            }
        }
    } else {
        const unsigned y = x[0] ^ x[1] ^ x[2] ^ x[3] ^ __float_as_uint(acc[0] + acc[1] + acc[2] + acc[3] + ss);
        if (y == 0x12345678u) out[blockIdx.x] = (__bf16)1.f;
    }
}

template <int F> static void launch(const __bf16* W, const __bf16* A, const __bf16* G, int K, int N, __bf16* out, const __bf16* R) {
    hipLaunchKernelGGL(feat_kernel<F>, dim3(N / 16), dim3(512), 0, 0, W, A, G, K, N, out, R);
}

int main() {
    const int shapes[][2] = {{12288, 4096}, {4096, 4096}, {22016, 4096}, {4096, 11008}};
    const int feats[] = {0, 1, 3, 7, 11, 15, 23, 31};
    __bf16 *A, *G, *out, *R;
    CK(hipMalloc(&A, 8 * 11008 * 2)); CK(hipMemset(A, 0, 8 * 11008 * 2)); CK(hipMalloc(&G, 11008 * 2)); CK(hipMemset(G, 0, 11008 * 2));
    CK(hipMalloc(&out, 8 * 32000 * 2)); CK(hipMalloc(&R, 8 * 32000 * 2)); CK(hipMemset(R, 0, 8 * 32000 * 2));
    for (auto& sh : shapes) {
        const long rows = sh[0], K = sh[1], bytes = rows * K * 2;
        const int ncopy = (int)(1200000000L / bytes) < 2 ? 2 : (int)(1200000000L / bytes);
        __bf16* W; CK(hipMalloc(&W, bytes * ncopy)); CK(hipMemset(W, 0, bytes * ncopy));
        for (int f : feats) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int iters = 60;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) {
                    const __bf16* p = W + (long)(i % ncopy) * rows * K;
                    switch (f) {
                        case 0: launch<0>(p, A, G, K, rows, out, R); break;   case 1: launch<1>(p, A, G, K, rows, out, R); break;
                        case 3: launch<3>(p, A, G, K, rows, out, R); break;   case 7: launch<7>(p, A, G, K, rows, out, R); break;
                        case 11: launch<11>(p, A, G, K, rows, out, R); break; case 15: launch<15>(p, A, G, K, rows, out, R); break;
                        case 23: launch<23>(p, A, G, K, rows, out, R); break; default: launch<31>(p, A, G, K, rows, out, R); break;
                    }
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("[%6ld x %5ld] %6.1f MB  features %2d (%s%s%s%s%s): %6.1f us = %5.2f TB/s\n", rows, K, bytes / 1e6, f, f & 1 ? "A " : "", f & 2 ? "mfma " : "",
                   f & 4 ? "reduce+store " : "", f & 8 ? "norm " : "", f & 16 ? "residual" : "", ms * 1e3 / iters, bytes / (ms * 1e-3 / iters) / 1e12);
        }
        CK(hipFree(W));
    }
    return 0;
}
