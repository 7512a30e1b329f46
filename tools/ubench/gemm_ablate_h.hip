// Ablation of the 256x256 / 16-wave GEMM main loop (default M=N=K=4096: 256 tiles, one per CU).  Timing only, plain epilogue.
//   FLAGS 0 baseline (BK=64, 2 stages, __syncthreads)   1 no loads in the loop   2 no ds_read/mfma   4 no barrier
//   FLAGS 16: BK=32, 4-stage ring, counted vmcnt (2 K-tiles in flight), raw s_barrier
//   FLAGS 32: BK=64, 2 stages, but the next tile's loads are issued in two halves (before each k-step)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int TM = 256, TN = 256;

__device__ __forceinline__ void glds16(const bf16* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
// 128-byte rows: one instruction = 8 rows
__device__ __forceinline__ void stage8(const bf16* base, long ld, int row0, int k0, char* lds, int group, int lane) {
    const int rsub = lane >> 3;
    glds16(base + (long)(row0 + group * 8 + rsub) * ld + k0 + (((lane & 7) ^ rsub) << 3), lds + group * 1024);
}
__device__ __forceinline__ int swz64(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }
// 64-byte rows: one instruction = 16 rows
__device__ __forceinline__ void stage16(const bf16* base, long ld, int row0, int k0, char* lds, int group, int lane) {
    const int rsub = lane >> 2;
    glds16(base + (long)(row0 + group * 16 + rsub) * ld + k0 + (((lane & 3) ^ swz64(rsub)) << 3), lds + group * 1024);
}

template <int FLAGS>
__global__ __launch_bounds__(1024, 4) void gk(const bf16* A, const bf16* B, bf16* C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    if (FLAGS & 16) {
        constexpr int ST = 512 * 64;                     // 32 KiB per stage (BK = 32)
        const int nt = K / 32;
        auto stage = [&](int t, int buf) {
            char* a = smem + buf * ST; char* b = a + 256 * 64;
            stage16(A, K, m0, t * 32, a, wave, lane);     // 16 groups of 16 rows
            stage16(B, K, n0, t * 32, b, wave, lane);
        };
        stage(0, 0); stage(1, 1); stage(2, 2);
        const int frag = fr * 64 + ((fq ^ swz64(fr)) << 4);
        for (int t = 0; t < nt; ++t) {
            const int ahead = nt - 1 - t;
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (t + 3 < nt) stage(t + 3, (t + 3) & 3);
            const char* a = smem + (t & 3) * ST + (wm * 64) * 64 + frag;
            const char* b = smem + (t & 3) * ST + 256 * 64 + (wn * 64) * 64 + frag;
            bf16x8 xa[4], wb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { xa[i] = *(const bf16x8*)(a + i * 1024); wb[i] = *(const bf16x8*)(b + i * 1024); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
    } else {
        constexpr int ST = 512 * 128;                    // 64 KiB per stage (BK = 64)
        const int nt = K / 64;
        auto stage_half = [&](int t, int buf, int h) {   // h = 0: A rows, 1: B rows (2 instructions per wave each)
            char* base = smem + buf * ST + h * 256 * 128;
            const bf16* src = h ? B : A;
            const int r0 = h ? n0 : m0;
            stage8(src, K, r0, t * 64, base, wave, lane);
            stage8(src, K, r0, t * 64, base, 16 + wave, lane);
        };
        auto compute_ks = [&](const char* a_lds, const char* b_lds, int ks) {
            bf16x8 xa[4], wb[4];
            const int chunk = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int r = wm * 64 + i * 16 + fr; xa[i] = *(const bf16x8*)(a_lds + r * 128 + ((chunk ^ (r & 7)) << 4)); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int r = wn * 64 + j * 16 + fr; wb[j] = *(const bf16x8*)(b_lds + r * 128 + ((chunk ^ (r & 7)) << 4)); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        };
        stage_half(0, 0, 0); stage_half(0, 0, 1);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            const char* a_lds = smem + cur * ST;
            const char* b_lds = a_lds + 256 * 128;
            const bool ld = !(FLAGS & 1) && t + 1 < nt;
            if (ld) { stage_half(t + 1, cur ^ 1, 0); if (!(FLAGS & 32)) stage_half(t + 1, cur ^ 1, 1); }
            if (!(FLAGS & 2)) compute_ks(a_lds, b_lds, 0);
            if (ld && (FLAGS & 32)) stage_half(t + 1, cur ^ 1, 1);
            if (!(FLAGS & 2)) compute_ks(a_lds, b_lds, 1);
            if (!(FLAGS & 4)) __syncthreads();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            bf16* cp = C + (long)(m0 + wm * 64 + i * 16 + fr) * N + n0 + wn * 64 + j * 16 + fq * 4;
            for (int e = 0; e < 4; ++e) cp[e] = (bf16)acc[i][j][e];
        }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int FLAGS> void run(const char* name, const bf16* A, const bf16* B, bf16* C, int M, int N, int K) {
    const int lds = 128 * 1024;
    CK(hipFuncSetAttribute((const void*)gk<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int tiles = (M / TM) * (N / TN);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gk<FLAGS>), dim3(tiles), dim3(1024), lds, 0, A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((gk<FLAGS>), dim3(tiles), dim3(1024), lds, 0, A, B, C, M, N, K);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
    printf("%-52s %8.1f us  %8.1f TF/s-equivalent\n", name, ms * 1000, 2.0 * M * N * K / ms / 1e9);
}

template <int FLAGS>
__global__ __launch_bounds__(512, 2) void gk8(const bf16* A, const bf16* B, bf16* C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    constexpr int ST = 512 * 128;
    const int nt = K / 64;
    auto stage = [&](int t, int buf) {
        char* a = smem + buf * ST; char* b = a + 256 * 128;
#pragma unroll
        for (int p = 0; p < 4; ++p) { stage8(A, K, m0, t * 64, a, p * 8 + wave, lane); stage8(B, K, n0, t * 64, b, p * 8 + wave, lane); }
    };
    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (!(FLAGS & 1) && t + 1 < nt) stage(t + 1, cur ^ 1);
        if (!(FLAGS & 2)) {
            const char* a_lds = smem + cur * ST + (wm * 128 + fr) * 128;
            const char* b_lds = smem + cur * ST + 256 * 128 + (wn * 64 + fr) * 128;
            const int c0 = (fq ^ (fr & 7)) << 4, c1 = ((4 + fq) ^ (fr & 7)) << 4;
            bf16x8 xa0[8], wb0[4], xa1[8], wb1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wb0[j] = *(const bf16x8*)(b_lds + j * 2048 + c0);
#pragma unroll
            for (int i = 0; i < 8; ++i) xa0[i] = *(const bf16x8*)(a_lds + i * 2048 + c0);
            if (FLAGS & 64) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wb1[j] = *(const bf16x8*)(b_lds + j * 2048 + c1);
#pragma unroll
                for (int i = 0; i < 8; ++i) xa1[i] = *(const bf16x8*)(a_lds + i * 2048 + c1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb0[j], xa0[i], acc[i][j], 0, 0, 0);
            if (!(FLAGS & 64)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wb1[j] = *(const bf16x8*)(b_lds + j * 2048 + c1);
#pragma unroll
                for (int i = 0; i < 8; ++i) xa1[i] = *(const bf16x8*)(a_lds + i * 2048 + c1);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb1[j], xa1[i], acc[i][j], 0, 0, 0);
        }
        if (!(FLAGS & 4)) __syncthreads();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) {
            bf16* cp = C + (long)(m0 + wm * 128 + i * 16 + fr) * N + n0 + wn * 64 + j * 16 + fq * 4;
            for (int e = 0; e < 4; ++e) cp[e] = (bf16)acc[i][j][e];
        }
}
template <int FLAGS> void run8(const char* name, const bf16* A, const bf16* B, bf16* C, int M, int N, int K) {
    const int lds = 128 * 1024;
    CK(hipFuncSetAttribute((const void*)gk8<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int tiles = (M / TM) * (N / TN);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gk8<FLAGS>), dim3(tiles), dim3(512), lds, 0, A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((gk8<FLAGS>), dim3(tiles), dim3(512), lds, 0, A, B, C, M, N, K);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
    printf("%-52s %8.1f us  %8.1f TF/s-equivalent\n", name, ms * 1000, 2.0 * M * N * K / ms / 1e9);
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    bf16 *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    const size_t nmax = (size_t)(M > N ? M : N) * K;
    unsigned short* h = (unsigned short*)malloc(nmax * 2);
    srand(1); for (size_t i = 0; i < nmax; ++i) h[i] = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
    CK(hipMemcpy(A, h, (size_t)M * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h, (size_t)N * K * 2, hipMemcpyHostToDevice));
    printf("M=%d N=%d K=%d (256x256 tile, 16 waves)\n", M, N, K);
    run<0>("baseline (BK=64, 2 stages, syncthreads)", A, B, C, M, N, K);
    run<1>("no loads in loop", A, B, C, M, N, K);
    run<2>("no ds_read/mfma (loads+barrier only)", A, B, C, M, N, K);
    run<4>("no barrier (vmcnt(0) only)", A, B, C, M, N, K);
    run<5>("no loads, no barrier (ds_read+mfma only)", A, B, C, M, N, K);
    run<32>("B half of the loads issued between the two k-steps", A, B, C, M, N, K);
    run<16>("BK=32, 4-stage ring, counted vmcnt, raw barrier", A, B, C, M, N, K);
    printf("-- 8 waves x 128x64 per wave (BK=64, 2 stages)\n");
    run8<0>("baseline", A, B, C, M, N, K);
    run8<64>("baseline, k-step-1 fragments read before k-step-0 mfma", A, B, C, M, N, K);
    run8<1>("no loads in loop", A, B, C, M, N, K);
    run8<2>("no ds_read/mfma (loads+barrier only)", A, B, C, M, N, K);
    run8<5>("no loads, no barrier (ds_read+mfma only)", A, B, C, M, N, K);
    run8<69>("ds_read+mfma only, early k-step-1 reads", A, B, C, M, N, K);
    return 0;
}
