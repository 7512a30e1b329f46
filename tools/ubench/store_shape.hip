// What does the epilogue of the persistent 256x256 GEMM pay for its C stores, as a function of the per-instruction address shape?
// One 256-thread workgroup per CU (4 waves, one per SIMD, as gemm_bf16_wp_kernel), each walking 256x256 bf16 output tiles of a
// [M, N] matrix (row stride N); a wave owns a 128x128 quadrant = 32 stores of 16 bytes per lane.  Between tiles the wave runs KSTEPS x 128
// register-only MFMAs (the K loop's matrix work, no memory traffic), so stores of one tile can drain under the next tile's MFMAs exactly as in
// the real kernel.  Shapes of ONE wave store instruction:
//   0: lane (fr, fq) -> row 16 i + fr, columns 32 p + 8 fq .. +7        = 16 rows x 64 B   (what the kernel does today: weights as MFMA A operand)
//   1: lane (fr, fq) -> row 16 i + 4 fq + r, columns 8 fr .. +7          =  4 rows x 256 B  (activations as MFMA A operand, weight row 8 fr + J)
//   2: lane l       -> row 4 k + (l >> 4), columns 8 (l & 15) .. +7     =  4 rows x 256 B, consecutive rows (an LDS-transposed epilogue)
// Prints us per tile and the store cost = time(with stores) - time(MFMAs only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SHAPE, bool STORE, bool NT>
__global__ __launch_bounds__(256, 1) void store_kernel(unsigned short* __restrict__ C, int M, int N, int ksteps, float* sink) {
    extern __shared__ char smem[];                    // 128 KiB requested: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = M / 256, tiles_n = N / 256, ntiles = tiles_m * tiles_n;
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane ^ e)); }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        for (int s = 0; s < ksteps; ++s) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        if (STORE) {
            const int tm = t / tiles_n, tn = t % tiles_n;
            unsigned short* base = C + ((long)tm * 256 + wr * 128) * N + tn * 256 + wc * 128;
            u32x4 v;
            v[0] = __float_as_uint(acc[0][0]); v[1] = __float_as_uint(acc[1][1]); v[2] = __float_as_uint(acc[2][2]); v[3] = __float_as_uint(acc[3][3]);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                long off;
                if (SHAPE == 0) { const int i = k >> 2, p = k & 3; off = (long)(16 * i + fr) * N + 32 * p + 8 * fq; }
                else if (SHAPE == 1) { const int i = k >> 2, r = k & 3; off = (long)(16 * i + 4 * fq + r) * N + 8 * fr; }
                else { off = (long)(4 * k + (lane >> 4)) * N + 8 * (lane & 15); }
                u32x4* p = (u32x4*)(base + off);
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            }
        }
    }
    float x = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) x += acc[i][0];
    if (x == 123.456f) sink[0] = x;
}

template <int SHAPE, bool STORE, bool NT>
static float run(unsigned short* C, int M, int N, int ksteps, float* sink, int iters) {
    CK(hipFuncSetAttribute((const void*)store_kernel<SHAPE, STORE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((store_kernel<SHAPE, STORE, NT>), dim3(256), dim3(256), 128 * 1024, 0, C, M, N, ksteps, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / iters < best) best = ms / iters;
    }
    return best * 1000.f;      // us per launch
}

int main() {
    const int M = 393984, N = 2304;                    // CLIP qkv output at B = 16 (rounded down to whole tiles): 1.8 GB
    unsigned short* C; CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMemset(C, 0, (size_t)M * N * 2));
    float* sink; CK(hipMalloc(&sink, 64));
    const int ntiles = (M / 256) * (N / 256);
    const float rounds = ntiles / 256.f;
    for (int ks : {0, 12, 48}) {
        const float base = ks ? run<0, false, false>(C, M, N, ks, sink, 3) : 0.f;
        printf("K-steps of matrix work per tile: %d   (MFMAs only: %.1f us per launch, %.2f us per tile)\n", ks, base, base / rounds);
        const float t0 = run<0, true, false>(C, M, N, ks, sink, 3), t1 = run<1, true, false>(C, M, N, ks, sink, 3), t2 = run<2, true, false>(C, M, N, ks, sink, 3);
        const float n0 = run<0, true, true>(C, M, N, ks, sink, 3), n1 = run<1, true, true>(C, M, N, ks, sink, 3);
        const double gb = (double)M * N * 2 / 1e9;
        printf("  shape 0 (16 rows x 64 B)        %8.1f us  store cost %6.2f us/tile  %5.2f TB/s\n", t0, (t0 - base) / rounds, gb / t0 * 1e3 / 1e3);
        printf("  shape 1 (4 rows x 256 B, r+4fq) %8.1f us  store cost %6.2f us/tile  %5.2f TB/s\n", t1, (t1 - base) / rounds, gb / t1 * 1e3 / 1e3);
        printf("  shape 2 (4 consecutive rows)    %8.1f us  store cost %6.2f us/tile  %5.2f TB/s\n", t2, (t2 - base) / rounds, gb / t2 * 1e3 / 1e3);
        printf("  shape 0, non-temporal           %8.1f us  store cost %6.2f us/tile\n", n0, (n0 - base) / rounds);
        printf("  shape 1, non-temporal           %8.1f us  store cost %6.2f us/tile\n", n1, (n1 - base) / rounds);
    }
    return 0;
}
