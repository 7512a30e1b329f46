// ds_read_b128 bank-conflict probe: cycles per wave-instruction for the fragment address patterns the GEMM kernels use.
//   hipcc -O3 --offload-arch=gfx950 -o lds_conflict lds_conflict.hip && ./lds_conflict
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ int addr_of(int pat, int lane, int it) {
    const int k = it & 3;
    switch (pat) {
        case 0: { const int r = lane & 15, c = (k & 1) * 4 + (lane >> 4); return r * 128 + ((c ^ (r & 7)) << 4); }          // 16x16x32, 128-B rows, ^ (row&7)
        case 1: { const int r = lane & 31, c = k * 2 + (lane >> 5); return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }        // 32x32x16, 128-B rows, ^ ((row>>1)&7)
        case 2: { const int r = lane & 31, c = k * 2 + (lane >> 5); return r * 128 + ((c ^ (r & 7)) << 4); }               // 32x32x16, 128-B rows, ^ (row&7)
        case 3: { const int r = lane & 31, c = k * 2 + (lane >> 5); return r * 128 + (c << 4); }                            // 32x32x16, linear
        case 4: { const int r = lane & 31, c = (k & 1) * 2 + (lane >> 5); return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }   // 32x32x16, 64-B rows, ^ ((row>>2)&3)
        case 5: return lane * 16;                                                                                           // contiguous
        case 6: { const int r = lane & 31, c = k * 2 + (lane >> 5); return r * 128 + ((c ^ ((r >> 2) & 7)) << 4); }        // 32x32x16, ^ ((row>>2)&7)
        case 7: { const int r = lane & 31, c = k * 2 + (lane >> 5); return r * 128 + ((c ^ (((r >> 1) & 3) | ((r >> 2) & 4))) << 4); }
        default: { const int r = lane & 15, c = (k & 1) * 4 + (lane >> 4); return r * 128 + (c << 4); }                     // 16x16x32 linear
    }
}

__global__ __launch_bounds__(256) void probe(int pat, int iters, long long* out, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((unsigned*)lds)[i] = i;
    __syncthreads();
    int a[4];
    for (int k = 0; k < 4; ++k) a[k] = addr_of(pat, lane, k) + (threadIdx.x >> 6) * 4096 * 2;
    u32x4 s = {0, 0, 0, 0};
    const int lbase = (int)(size_t)(__attribute__((address_space(3))) char*)lds;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            u32x4 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lbase + a[u & 3] + ((u >> 2) & 1) * 4096));
            asm volatile("" : "+v"(v));
            if (u == 15) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); s += v; }
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s[0] + s[1] + s[2] + s[3] == 0x12345) sink[0] = s[0];
}

int main() {
    long long* out; unsigned* sink;
    CK(hipMalloc(&out, 8)); CK(hipMalloc(&sink, 4));
    const char* names[] = {"16x16x32 rows128 ^(row&7)", "32x32x16 rows128 ^((row>>1)&7)", "32x32x16 rows128 ^(row&7)", "32x32x16 rows128 linear",
                           "32x32x16 rows64 ^((row>>2)&3)", "contiguous lane*16", "32x32x16 rows128 ^((row>>2)&7)", "32x32x16 rows128 mixed", "16x16x32 rows128 linear"};
    const int iters = 2000;
    for (int p = 0; p < 9; ++p) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, p, iters, out, sink);
        CK(hipDeviceSynchronize());
        long long h; CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
        printf("%-34s %6.2f clk per ds_read_b128 per wave (4 waves/CU -> %5.1f B/clk/CU)\n", names[p], (double)h / (iters * 16.0), 4 * 1024.0 / ((double)h / (iters * 16.0)));
    }
    return 0;
}
