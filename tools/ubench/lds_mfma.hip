// LDS-read + MFMA loops with the exact fragment addressing of the GEMM kernels (no global traffic, no barriers).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ int swz64(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

// MODE 0: 128-B rows, XOR (row&7) swizzle, 64x64 per wave, K-tile 64 (2 k-steps)  [gemm_bf16_kernel]
// MODE 1: same without swizzle
// MODE 2: 64-B rows, F swizzle, 128x64 per wave, K-tile 32                         [gemm_bf16_x_kernel]
// MODE 3: 64-B rows, no swizzle, 128x64
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const bf16* in, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 16; i += blockDim.x) ((uint4*)smem)[i] = ((const uint4*)in)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    float s = 0;
    if (MODE < 2) {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
        const int wm = (w >> 1) & 1, wn = w & 1;
        for (int it = 0; it < iters; ++it) {
            const char* a_lds = smem + (it & 1) * 32768;
            const char* b_lds = a_lds + 16384;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 xa[4], wb[4];
                const int chunk = ks * 4 + fq;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int r = wm * 64 + i * 16 + fr; xa[i] = *(const bf16x8*)(a_lds + r * 128 + (((MODE == 0 ? (chunk ^ (r & 7)) : chunk)) << 4)); }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int r = wn * 64 + j * 16 + fr; wb[j] = *(const bf16x8*)(b_lds + r * 128 + (((MODE == 0 ? (chunk ^ (r & 7)) : chunk)) << 4)); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    } else {
        f32x4 acc[8][4];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
        const int wm = (w >> 2) & 1, wn = w & 3;
        const int frag_off = fr * 64 + (((MODE == 2 ? (fq ^ swz64(fr)) : fq)) << 4);
        for (int it = 0; it < iters; ++it) {
            const char* a_lds = smem + (it & 1) * 32768 + (wm * 128) * 64 + frag_off;
            const char* b_lds = smem + (it & 1) * 32768 + 16384 + (wn * 64) * 64 + frag_off;
            bf16x8 wb[4], xa[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) wb[j] = *(const bf16x8*)(b_lds + j * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i) xa[i] = *(const bf16x8*)(a_lds + i * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <typename F> double timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}
template <int MODE> void run(const char* name, const bf16* in, float* out, int threads, int blocks_per_cu) {
    CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    double ms = timeit([&] { hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 65536, 0, in, out, iters); });
    const double mf = MODE < 2 ? 32.0 : 32.0;      // MFMAs per wave per iteration
    printf("%-52s %d thr x %d blk/CU: %8.1f TF/s\n", name, threads, blocks_per_cu, 2.0 * 16 * 16 * 32 * mf * iters * blocks * (threads / 64) / ms / 1e9);
}
int main() {
    bf16* in; float* out;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 256 * 8 * 512 * 4));
    unsigned short* h = (unsigned short*)malloc(1 << 20);
    srand(1); for (int i = 0; i < (1 << 19); ++i) h[i] = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15);
    CK(hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice));
    run<0>("128B rows xor-swizzle, 64x64/wave (16 rd : 32 mfma)", in, out, 256, 1);
    run<0>("128B rows xor-swizzle, 64x64/wave (16 rd : 32 mfma)", in, out, 256, 2);
    run<0>("128B rows xor-swizzle, 64x64/wave (16 rd : 32 mfma)", in, out, 512, 1);
    run<1>("128B rows linear,      64x64/wave", in, out, 256, 2);
    run<2>("64B rows F-swizzle,   128x64/wave (12 rd : 32 mfma)", in, out, 256, 1);
    run<2>("64B rows F-swizzle,   128x64/wave (12 rd : 32 mfma)", in, out, 256, 2);
    run<2>("64B rows F-swizzle,   128x64/wave (12 rd : 32 mfma)", in, out, 512, 1);
    run<3>("64B rows linear,      128x64/wave", in, out, 512, 1);
    return 0;
}
