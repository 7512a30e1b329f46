// Ablation of the 128x128x64 GEMM main loop (default M=2048,N=4096,K=4096: 512 tiles, 2 blocks/CU).  Timing only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int BM = 128, BN = 128, BK = 64, TILE_BYTES = BM * BK * 2;

__device__ __forceinline__ void stage_tile(const bf16* __restrict__ base, long ld, int row0, int k0, char* lds, int wave, int lane) {
    const int rsub = lane >> 3, chunk_src = (lane & 7) ^ rsub;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const bf16* src = base + (long)(row0 + p * 32 + wave * 8 + rsub) * ld + k0 + chunk_src * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + (p * 32 + wave * 8) * 128), 16, 0, 0);
    }
}
// FLAGS: 1 = no loads in loop, 2 = no mfma/ds_read, 4 = no barrier, 8 = loads issued after first k-step,
//        16 = 3-stage ring counted vmcnt (96 KB, 1 block/CU)
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void gk(const bf16* A, const bf16* B, bf16* C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = M / BM;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    const int nt = K / BK;
    auto stage = [&](int t, int buf) {
        char* a_lds = smem + buf * (2 * TILE_BYTES);
        stage_tile(A, K, m0, t * BK, a_lds, wave, lane);
        stage_tile(B, K, n0, t * BK, a_lds + TILE_BYTES, wave, lane);
    };
    auto compute_ks = [&](const char* a_lds, const char* b_lds, int ks) {
        const int fr = lane & 15, fq = lane >> 4;
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
    if (FLAGS & 16) {
        stage(0, 0); stage(1, 1);
        int buf = 0;
        for (int t = 0; t < nt; ++t) {
            if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (t + 2 < nt) stage(t + 2, buf >= 1 ? buf - 1 : 2);
            const char* a_lds = smem + buf * (2 * TILE_BYTES);
            compute_ks(a_lds, a_lds + TILE_BYTES, 0);
            compute_ks(a_lds, a_lds + TILE_BYTES, 1);
            buf = buf == 2 ? 0 : buf + 1;
        }
    } else {
        stage(0, 0);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            const char* a_lds = smem + cur * (2 * TILE_BYTES);
            if (!(FLAGS & 1) && !(FLAGS & 8) && t + 1 < nt) stage(t + 1, cur ^ 1);
            if (!(FLAGS & 2)) compute_ks(a_lds, a_lds + TILE_BYTES, 0);
            if (!(FLAGS & 1) && (FLAGS & 8) && t + 1 < nt) stage(t + 1, cur ^ 1);
            if (!(FLAGS & 2)) compute_ks(a_lds, a_lds + TILE_BYTES, 1);
            if (!(FLAGS & 4)) __syncthreads();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    const int fr = lane & 15, fq = lane >> 4;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            bf16* cp = C + (long)(m0 + wm * 64 + i * 16 + fr) * N + n0 + wn * 64 + j * 16 + fq * 4;
            for (int e = 0; e < 4; ++e) cp[e] = (bf16)acc[i][j][e];
        }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int FLAGS> void run(const char* name, const bf16* A, const bf16* B, bf16* C, int M, int N, int K) {
    const int lds = ((FLAGS & 16) ? 3 : 2) * 2 * TILE_BYTES;
    CK(hipFuncSetAttribute((const void*)gk<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int tiles = (M / BM) * (N / BN);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gk<FLAGS>), dim3(tiles), dim3(256), lds, 0, A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((gk<FLAGS>), dim3(tiles), dim3(256), lds, 0, A, B, C, M, N, K);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
    printf("%-44s %8.1f us  %8.1f TF/s-equivalent\n", name, ms * 1000, 2.0 * M * N * K / ms / 1e9);
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 2048, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    bf16 *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    const size_t nmax = (size_t)(M > N ? M : N) * K;
    unsigned short* h = (unsigned short*)malloc(nmax * 2);
    srand(1); for (size_t i = 0; i < nmax; ++i) h[i] = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
    CK(hipMemcpy(A, h, (size_t)M * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h, (size_t)N * K * 2, hipMemcpyHostToDevice));
    printf("M=%d N=%d K=%d\n", M, N, K);
    run<0>("baseline (2-stage, syncthreads)", A, B, C, M, N, K);
    run<1>("no loads in loop", A, B, C, M, N, K);
    run<2>("no ds_read/mfma (loads+barrier only)", A, B, C, M, N, K);
    run<4>("no barrier (vmcnt(0) only)", A, B, C, M, N, K);
    run<5>("no loads, no barrier (ds_read+mfma only)", A, B, C, M, N, K);
    run<8>("loads issued after first k-step", A, B, C, M, N, K);
    run<16>("3-stage ring, counted vmcnt, raw barrier", A, B, C, M, N, K);
    return 0;
}
