// How fast can a decode projection's weight stream be read, as a function of the per-instruction address pattern?
//   pattern 0: what an MFMA B-operand load needs: lane (fr, fq) reads 16 bytes of row n0+fr at k = 8 fq  -> 16 rows x 64 B per wave instruction
//   pattern 1: the same 16 x K x 2-byte block of a workgroup read contiguously: 1 KiB per wave instruction
// Same grid (rows/16 workgroups x 8 waves), same bytes per wave, 8 loads in flight per wave.  Prints TB/s over a rotation of buffers > L2+MALL.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PAT>
__global__ __launch_bounds__(512) void stream_kernel(const unsigned short* __restrict__ W, int K, unsigned* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const long n0 = (long)blockIdx.x * 16;
    const int kw = K / 8, nk = kw / 32;
    u32x4 acc = {0, 0, 0, 0};
    if (PAT == 0) {
        const unsigned short* bp = W + (n0 + fr) * K + (long)w * kw + fq * 8;
        for (int s = 0; s < nk; s += 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = s + u < nk ? *(const u32x4*)(bp + (s + u) * 32) : (u32x4){0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
    } else {
        // block = 16 rows * K * 2 bytes contiguous; wave w takes 1 KiB pieces w, w+8, ...
        const char* base = (const char*)(W + n0 * K);
        const int pieces = 16 * K * 2 / 1024 / 8;          // per wave
        for (int s = 0; s < pieces; s += 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = s + u < pieces ? *(const u32x4*)(base + ((long)((s + u) * 8 + w) * 64 + lane) * 16) : (u32x4){0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
    }
    const unsigned x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (x == 0x12345678u) out[blockIdx.x] = x;             // keeps the loads alive; practically never true
}

int main() {
    const int shapes[][2] = {{12288, 4096}, {4096, 4096}, {22016, 4096}, {4096, 11008}, {32000, 4096}};
    unsigned* out; CK(hipMalloc(&out, 1 << 20));
    for (auto& sh : shapes) {
        const long rows = sh[0], K = sh[1], bytes = rows * K * 2;
        const int ncopy = (int)(1200000000L / bytes) < 2 ? 2 : (int)(1200000000L / bytes);
        unsigned short* W; CK(hipMalloc(&W, bytes * ncopy)); CK(hipMemset(W, 1, bytes * ncopy));
        for (int pat = 0; pat < 2; ++pat) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int iters = 60;
            for (int rep = 0; rep < 2; ++rep) {              // first repetition warms clocks
                CK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) {
                    const unsigned short* p = W + (long)(i % ncopy) * rows * K;
                    if (pat == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(rows / 16), dim3(512), 0, 0, p, (int)K, out);
                    else hipLaunchKernelGGL(stream_kernel<1>, dim3(rows / 16), dim3(512), 0, 0, p, (int)K, out);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("[%6ld x %5ld] %6.1f MB  pattern %d (%s): %6.1f us/launch = %5.2f TB/s\n", rows, K, bytes / 1e6, pat,
                   pat ? "1 KiB contiguous per wave instruction" : "16 rows x 64 B per wave instruction (MFMA operand layout)", ms * 1e3 / iters,
                   bytes / (ms * 1e-3 / iters) / 1e12);
        }
        CK(hipFree(W));
    }
    return 0;
}
