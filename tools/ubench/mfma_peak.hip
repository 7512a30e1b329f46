// Microbenchmarks: (1) register-only MFMA loop, (2) LDS-read + MFMA loop shaped like the GEMM inner k-step.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256) void reg_only(const bf16* in, float* out, int iters) {
    bf16x8 a = *(const bf16x8*)(in + threadIdx.x * 8), b = *(const bf16x8*)(in + 4096 + threadIdx.x * 8);
    if (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < 16; ++i) s += acc[i][0];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

// LDS-fed: per iteration 8 ds_read_b128 + 16 MFMA (64x64 per wave, like the 128x128 kernel) or 12 reads + 32 MFMA (128x64)
template <int MT, int NT_>
__global__ void lds_fed(const bf16* in, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 32768 / 16; i += blockDim.x) ((uint4*)smem)[i] = ((const uint4*)in)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[MT][NT_];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT_; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    const char* base = smem + ((w * 997) & 8191);
    const int off = fr * 64 + (fq << 4);
    for (int it = 0; it < iters; ++it) {
        bf16x8 xa[MT], wb[NT_];
        const char* p = base + ((it & 7) << 10) + off;
#pragma unroll
        for (int j = 0; j < NT_; ++j) wb[j] = *(const bf16x8*)(p + j * 1024);
#pragma unroll
        for (int i = 0; i < MT; ++i) xa[i] = *(const bf16x8*)(p + 8192 + i * 1024);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT_; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT_; ++j) s += acc[i][j][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <typename F> double timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}
int main() {
    bf16* in; float* out;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 256 * 8 * 512 * 4));
    unsigned short* h = (unsigned short*)malloc(1 << 20);
    srand(1); for (int i = 0; i < (1 << 19); ++i) h[i] = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15);
    CK(hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice));
    const int iters = 20000;
    for (int wpc : {4, 8}) {
        int blocks = 256 * wpc / 4;
        double ms = timeit([&] { hipLaunchKernelGGL((reg_only<16>), dim3(blocks), dim3(256), 0, 0, in, out, iters); });
        printf("reg-only 16x16x32  %d waves/CU: %.1f TF/s\n", wpc, 2.0 * 16 * 16 * 32 * 16 * iters * blocks * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((reg_only<32>), dim3(blocks), dim3(256), 0, 0, in, out, iters); });
        printf("reg-only 32x32x16  %d waves/CU: %.1f TF/s\n", wpc, 2.0 * 32 * 32 * 16 * 4 * iters * blocks * 4 / ms / 1e9);
    }
    CK(hipFuncSetAttribute((const void*)lds_fed<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)lds_fed<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int wpc : {4, 8}) {
        int threads = 256, blocks = 256 * wpc / 4;
        double ms = timeit([&] { hipLaunchKernelGGL((lds_fed<4, 4>), dim3(blocks), dim3(threads), 32768, 0, in, out, iters / 4); });
        printf("lds-fed 64x64/wave (8 rd:16 mfma)  %d waves/CU: %.1f TF/s\n", wpc, 2.0 * 16 * 16 * 32 * 16 * (iters / 4) * blocks * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((lds_fed<8, 4>), dim3(blocks), dim3(threads), 32768, 0, in, out, iters / 4); });
        printf("lds-fed 128x64/wave (12 rd:32 mfma) %d waves/CU: %.1f TF/s\n", wpc, 2.0 * 16 * 16 * 32 * 32 * (iters / 4) * blocks * 4 / ms / 1e9);
    }
    return 0;
}
