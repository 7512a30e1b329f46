// 256x256 tile, 8 waves in two groups of four that run half a phase apart ("ping-pong"): while one group issues its 16 MFMAs
// the other issues its ds_reads and the next global->LDS half-tile.  BK = 64, two 64-KiB K-tile buffers of four 16-KiB
// half-tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255), one half-tile restaged per phase, 4 half-tiles in flight
// (counted vmcnt, never 0 in the main loop).  Checked against a plain 16-wave kernel, then timed.
//   hipcc -O3 --offload-arch=gfx950 -o gemm_pingpong gemm_pingpong.hip && ./gemm_pingpong [M N K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <type_traits>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
constexpr int TM = 256, TN = 256;

__device__ __forceinline__ void glds16(const bf16* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

__device__ long long g_clk[2];
template <int V> using IC = std::integral_constant<int, V>;

// Hazards, in barrier intervals I_n (group 0: load part of phase g in I_2g, MFMA part in I_2g+1; group 1 one interval later):
//  * a half-tile read in phase g is restaged in phase >= g+2 (both groups' reads have retired behind a barrier by then);
//  * the wait in phase g (after that phase's issue) covers the half-tiles first read in phase g+1, and a barrier passed by
//    every wave lies between the two.
// Issue order of half-tiles: A0,B0,B1,A1 of tile 0, then of tile 1, ...; phase g issues sequence number g+6.
// FLAGS: 1 setprio around the MFMA part; ablations: 2 no global loads in the loop, 4 no ds_reads in the loop, 8 no stagger,
// 16 no MFMAs, 32 no barriers in the loop (only valid with 2+4+8)
template <int FLAGS>
__global__ __launch_bounds__(512, 2) void gpp(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    const int nt = K / 64;

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: 16 wave-instructions of 8 rows per half-tile, two per wave
    const int rsub = lane >> 3;
    const long sw = ((lane & 7) ^ rsub) << 3;
    const bf16* gA = A + (long)(m0 + wave * 16 + rsub) * K + sw;
    const bf16* gB = B + (long)(n0 + wave * 16 + rsub) * K + sw;
    auto issue_raw = [&](int tile, int kind, int boff) {     // kind 0:A0 1:B0 2:B1 3:A1; boff = LDS byte offset of the K-tile buffer
        const int isB = (kind == 1 || kind == 2), h = (kind >= 2);
        const bf16* src = (isB ? gB : gA) + (long)h * 128 * K + (long)tile * 64;
        char* dst = smem + boff + isB * 32768 + h * 16384 + wave * 2048;
        glds16(src, dst);
        glds16(src + 8L * K, dst + 1024);
    };
    auto issue = [&](int tile, int kind, int boff) { if (!(FLAGS & 2) && tile < nt) issue_raw(tile, kind, boff); };
    const int c0 = (fq ^ (fr & 7)) << 4, c1 = ((4 + fq) ^ (fr & 7)) << 4;
    const char* ldsA0 = smem + (64 * wr + fr) * 128 + c0;
    const char* ldsA1 = smem + (64 * wr + fr) * 128 + c1;
    const char* ldsB0 = smem + 32768 + (32 * wc + fr) * 128 + c0;
    const char* ldsB1 = smem + 32768 + (32 * wc + fr) * 128 + c1;
    bf16x8 xa[2][4], wb[2][2][2];
    auto rdA = [&](int boff, int ha, bool force = false) {
        if ((FLAGS & 4) && !force) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xa[0][i] = *(const bf16x8*)(ldsA0 + boff + ha * 16384 + i * 2048);
            xa[1][i] = *(const bf16x8*)(ldsA1 + boff + ha * 16384 + i * 2048);
        }
    };
    auto rdB = [&](int boff, int hb, bool force = false) {
        if ((FLAGS & 4) && !force) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wb[hb][0][j] = *(const bf16x8*)(ldsB0 + boff + hb * 16384 + j * 2048);
            wb[hb][1][j] = *(const bf16x8*)(ldsB1 + boff + hb * 16384 + j * 2048);
        }
    };
    auto mm = [&](int ha, int hb) {
        if (FLAGS & 16) return;
        if (FLAGS & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ha][hb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[hb][ks][j], xa[ks][i], acc[ha][hb][i][j], 0, 0, 0);
        if (FLAGS & 1) __builtin_amdgcn_s_setprio(0);
    };
#define PP_BAR() do { if (!(FLAGS & 32)) __builtin_amdgcn_s_barrier(); } while (0)
    // after phase g's issue, all but the newest min(4, 4nt-3-g) half-tiles must have landed (2 loads per thread each)
    auto wait_for = [&](int g) {
        const int rem = 4 * nt - 3 - g;
        if (rem >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (rem == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
#define PP_PHASE_END(ha, hb)                      \
        __builtin_amdgcn_sched_barrier(0);        \
        PP_BAR();                                 \
        __builtin_amdgcn_sched_barrier(0);        \
        mm(ha, hb);                               \
        __builtin_amdgcn_sched_barrier(0);        \
        PP_BAR();                                 \
        asm volatile("" ::: "memory");

    // prologue: tile 0 and the first two half-tiles of tile 1
    issue_raw(0, 0, 0); issue_raw(0, 1, 0); issue_raw(0, 2, 0); issue_raw(0, 3, 0); issue_raw(1, 0, 65536); issue_raw(1, 1, 65536);
    if (FLAGS & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (FLAGS & 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); rdA(0, 0, true); rdB(0, 0, true); rdB(0, 1, true); }
    if (wr == 1 && !(FLAGS & 8)) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind
    for (int t = 0; t < nt; ++t) {
        const int boff = (t & 1) << 16, noff = 65536 - boff;
        // phase 1: quadrant (0,0)
        rdB(boff, 0); rdA(boff, 0);
        issue(t + 1, 2, noff);
        wait_for(4 * t);
        PP_PHASE_END(0, 0)
        // phase 2: quadrant (0,1)
        rdB(boff, 1);
        issue(t + 1, 3, noff);
        wait_for(4 * t + 1);
        PP_PHASE_END(0, 1)
        // phase 3: quadrant (1,1)
        rdA(boff, 1);
        issue(t + 2, 0, boff);
        wait_for(4 * t + 2);
        PP_PHASE_END(1, 1)
        // phase 4: quadrant (1,0)
        issue(t + 2, 1, boff);
        wait_for(4 * t + 3);
        PP_PHASE_END(1, 0)
    }
    if (wr == 0 && !(FLAGS & 8)) __builtin_amdgcn_s_barrier();
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = clock64() - ck0; g_clk[1] = wall_clock64() - wk0; }

#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bf16* cp = C + (long)(m0 + 128 * ha + 64 * wr + i * 16 + fr) * N + n0 + 128 * hb + 32 * wc + j * 16 + fq * 4;
                    const f32x4 v = acc[ha][hb][i][j];
                    *(bf16x4*)cp = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                }
}


// ---------------------------------------------------------------------------------------------------------------------
// 4 waves (one per SIMD), 128x128 per wave, v_mfma_f32_32x32x16_bf16: half the LDS read bytes of the 16-wave kernel and a
// quarter of the MFMA operand fetches per flop.  BK = 64, two 64-KiB stages; 128-B rows, chunk ^ ((row >> 1) & 7) swizzle
// (conflict-free for the 32-row fragment reads).  FLAGS: 1 next slice's fragments read before the current slice's MFMAs
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int FLAGS>
__global__ __launch_bounds__(256, 1) void g4w(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    const int nt = K / 64;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // staging: 32 groups of 8 rows per operand, 8 per wave
    const int rsub = lane >> 3;
    auto stage = [&](int t, int boff) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int g = wave * 8 + q, r = g * 8 + rsub;
            const long sw = (long)(((lane & 7) ^ ((r >> 1) & 7)) << 3);
            glds16(A + (long)(m0 + r) * K + (long)t * 64 + sw, smem + boff + g * 1024);
            glds16(B + (long)(n0 + r) * K + (long)t * 64 + sw, smem + boff + 32768 + g * 1024);
        }
    };
    const int fsw = (l31 >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (((ks * 2 + hi) ^ fsw) << 4);
    const char* la = smem + (wr * 128 + l31) * 128;
    const char* lb = smem + 32768 + (wc * 128 + l31) * 128;
    bf16x8 fa[2][4], fb[2][4];
    auto rd = [&](int boff, int ks, int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[set][i] = *(const bf16x8*)(la + boff + koff[ks] + i * 4096);
            fb[set][i] = *(const bf16x8*)(lb + boff + koff[ks] + i * 4096);
        }
    };
    auto mm = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nt; ++t) {
        const int boff = (t & 1) << 16;
        if (t + 1 < nt) stage(t + 1, 65536 - boff);
        if (FLAGS & 1) {
            rd(boff, 0, 0);
            rd(boff, 1, 1); mm(0);
            rd(boff, 2, 0); mm(1);
            rd(boff, 3, 1); mm(0);
            mm(1);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { rd(boff, ks, 0); mm(0); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = clock64() - ck0; g_clk[1] = wall_clock64() - wk0; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16* cp = C + (long)(m0 + wr * 128 + i * 32 + l31) * N + n0 + wc * 128 + j * 32 + 8 * q + 4 * hi;
                *(bf16x4*)cp = (bf16x4){(bf16)acc[i][j][4 * q], (bf16)acc[i][j][4 * q + 1], (bf16)acc[i][j][4 * q + 2], (bf16)acc[i][j][4 * q + 3]};
            }
}

// ---------------------------------------------------------------------------------------------------------------------
// 4 waves, 128x128 per wave, 32x32x16 MFMA, K in units of 32 (A 256x32 + B 256x32 = 32 KiB) through a ring of 4 LDS slots:
// unit u+4 is requested right after the barrier that retires unit u, so 2-3 units (64-96 KiB) stay in flight across every
// barrier; one barrier per unit (32 MFMAs per wave).  Fragments are double-buffered in registers one 16-wide k-slice ahead and
// the ds_reads are pinned between the MFMAs with sched_group_barrier.  64-B rows, chunk ^ ((row >> 2) & 3) swizzle.
template <int FLAGS>
__global__ __launch_bounds__(256, 1) void g4r(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    const int nu = K / 32;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const long sw = (long)(((lane & 3) ^ ((lane >> 4) & 3)) << 3);
    const bf16* gA = A + (long)(m0 + wave * 64 + (lane >> 2)) * K + sw;
    const bf16* gB = B + (long)(n0 + wave * 64 + (lane >> 2)) * K + sw;
    auto stage = [&](int u) {
        char* dst = smem + (u & 3) * 32768 + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(gA + (long)q * 16 * K + (long)u * 32, dst + q * 1024);
            glds16(gB + (long)q * 16 * K + (long)u * 32, dst + 16384 + q * 1024);
        }
    };
    const int fsw = (l31 >> 2) & 3;
    const int kc0 = ((hi ^ fsw) << 4), kc1 = (((2 + hi) ^ fsw) << 4);
    const char* la = smem + (wr * 128 + l31) * 64;
    const char* lb = smem + 16384 + (wc * 128 + l31) * 64;
    bf16x8 fa[2][4], fb[2][4];
    auto rd = [&](int u, int kl, int set) {
        const int off = (u & 3) * 32768 + (kl ? kc1 : kc0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[set][i] = *(const bf16x8*)(la + off + i * 2048);
            fb[set][i] = *(const bf16x8*)(lb + off + i * 2048);
        }
    };
    auto mm = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };
    auto pin = [&]() {                                       // 8 x {1 MFMA, 1 ds_read}, then the remaining 8 MFMAs
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    };
    stage(0); stage(1); stage(2); stage(3);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rd(0, 0, 0);
    for (int u = 0; u < nu; ++u) {
        // slice 2u: set 0; fetch slice 2u+1 (same unit) into set 1
        __builtin_amdgcn_sched_barrier(0);
        rd(u, 1, 1);
        mm(0);
        if (FLAGS & 1) pin();
        __builtin_amdgcn_sched_barrier(0);
        {   // unit u fully read by this wave; unit u+1 landed (units u+2, u+3 may still be in flight)
            const int rem = nu - 2 - u;
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (u + 4 < nu) stage(u + 4);
        // slice 2u+1: set 1; fetch slice 2u+2 (next unit) into set 0
        __builtin_amdgcn_sched_barrier(0);
        rd(u + 1, 0, 0);
        mm(1);
        if (FLAGS & 1) pin();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = clock64() - ck0; g_clk[1] = wall_clock64() - wk0; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16* cp = C + (long)(m0 + wr * 128 + i * 32 + l31) * N + n0 + wc * 128 + j * 32 + 8 * q + 4 * hi;
                *(bf16x4*)cp = (bf16x4){(bf16)acc[i][j][4 * q], (bf16)acc[i][j][4 * q + 1], (bf16)acc[i][j][4 * q + 2], (bf16)acc[i][j][4 * q + 3]};
            }
}

// Same ring as g4r with v_mfma_f32_16x16x32_bf16 (the shape the chip clocks higher on): 8x8 blocks of 16x16 per wave, one K unit
// of 32 = one MFMA k-slice; fragments double-buffered one unit ahead.
__device__ __forceinline__ int swz64(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }
template <int FLAGS>
__global__ __launch_bounds__(256, 1) void g4s(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    const int nu = K / 32;
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long sw = (long)(((lane & 3) ^ swz64(lane >> 2)) << 3);
    const bf16* gA = A + (long)(m0 + wave * 64 + (lane >> 2)) * K + sw;
    const bf16* gB = B + (long)(n0 + wave * 64 + (lane >> 2)) * K + sw;
    auto stage = [&](int u) {
        char* dst = smem + (u & 3) * 32768 + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(gA + (long)q * 16 * K + (long)u * 32, dst + q * 1024);
            glds16(gB + (long)q * 16 * K + (long)u * 32, dst + 16384 + q * 1024);
        }
    };
    const int kc = ((fq ^ swz64(fr)) << 4);
    const char* la = smem + (wr * 128 + fr) * 64 + kc;
    const char* lb = smem + 16384 + (wc * 128 + fr) * 64 + kc;
    bf16x8 fa[2][8], fb[2][8];
    auto rd = [&](int u, int set) {
        const int off = (u & 3) * 32768;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fa[set][i] = *(const bf16x8*)(la + off + i * 1024);
            fb[set][i] = *(const bf16x8*)(lb + off + i * 1024);
        }
    };
    auto mm = [&](int set) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };
    auto pin = [&]() {                                       // 16 x {2 MFMA, 1 ds_read}, then the remaining 32 MFMAs
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
    };
    stage(0); stage(1); stage(2); stage(3);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rd(0, 0);
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // invariant at the top of unit u: set (u&1) holds unit u's fragments; units u+1 landed and visible; slot of unit u is free
    auto unit = [&](int u, int set) {
        if (u + 4 < nu) stage(u + 4);
        __builtin_amdgcn_sched_barrier(0);
        rd(u + 1, set ^ 1);
        mm(set);
        if (FLAGS & 1) pin();
        __builtin_amdgcn_sched_barrier(0);
        {   // unit u+1 fully read by this wave; unit u+2 landed (u+3, u+4 may still be in flight)
            const int rem = nu - 3 - u;
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int u = 0; u < nu; u += 2) { unit(u, 0); unit(u + 1, 1); }
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = clock64() - ck0; g_clk[1] = wall_clock64() - wk0; }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bf16* cp = C + (long)(m0 + wr * 128 + i * 16 + fr) * N + n0 + wc * 128 + j * 16 + fq * 4;
            const f32x4 v = acc[i][j];
            *(bf16x4*)cp = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        }
}

// 4 waves x 128x128, 16x16x32 MFMA, BK=32 ring of 4: B fragments of the NEXT unit and the A fragments are streamed one per
// 8-MFMA row block (72 fragment VGPRs instead of 128), everything pinned with sched_group_barrier.
template <int FLAGS>
__global__ __launch_bounds__(256, 1) void g4t(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    const int nu = K / 32;
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long sw = (long)(((lane & 3) ^ swz64(lane >> 2)) << 3);
    const bf16* gA = A + (long)(m0 + wave * 64 + (lane >> 2)) * K + sw;
    const bf16* gB = B + (long)(n0 + wave * 64 + (lane >> 2)) * K + sw;
    auto stage = [&](int u) {
        char* dst = smem + (u & 3) * 32768 + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(gA + (long)q * 16 * K + (long)u * 32, dst + q * 1024);
            glds16(gB + (long)q * 16 * K + (long)u * 32, dst + 16384 + q * 1024);
        }
    };
    const int kc = ((fq ^ swz64(fr)) << 4);
    const char* la = smem + (wr * 128 + fr) * 64 + kc;
    const char* lb = smem + 16384 + (wc * 128 + fr) * 64 + kc;
    bf16x8 fa[2], fb[2][8];
    stage(0); stage(1); stage(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[0][j] = *(const bf16x8*)(lb + j * 1024);
    fa[0] = *(const bf16x8*)(la);
    auto unit = [&](int u, auto setc) {
        constexpr int set = decltype(setc)::value;
        if (FLAGS & 2) {                                       // branch-free body: past the end, unit nu-1 is re-staged into a free slot
            const int us = u + 3 < nu ? u + 3 : nu - 1;
            char* dst = smem + ((u + 3) & 3) * 32768 + wave * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                glds16(gA + (long)q * 16 * K + (long)us * 32, dst + q * 1024);
                glds16(gB + (long)q * 16 * K + (long)us * 32, dst + 16384 + q * 1024);
            }
        } else if (u + 3 < nu) stage(u + 3);
        const int o0 = (u & 3) * 32768, o1 = ((u + 1) & 3) * 32768;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < 7) fa[(i + 1) & 1] = *(const bf16x8*)(la + o0 + (i + 1) * 1024);
            else fa[0] = *(const bf16x8*)(la + o1);
            fb[set ^ 1][i] = *(const bf16x8*)(lb + o1 + i * 1024);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[set][j], fa[i & 1], acc[i][j], 0, 0, 0);
            if (FLAGS & 1) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if ((FLAGS & 2) || u + 3 < nu) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int u = 0; u < nu; u += 2) { unit(u, IC<0>{}); unit(u + 1, IC<1>{}); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = clock64() - ck0; g_clk[1] = wall_clock64() - wk0; }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bf16* cp = C + (long)(m0 + wr * 128 + i * 16 + fr) * N + n0 + wc * 128 + j * 16 + fq * 4;
            const f32x4 v = acc[i][j];
            *(bf16x4*)cp = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        }
}

// 4 waves x 128x128 with the K loop written instruction by instruction (inline asm): accumulators pinned to AGPRs ("+a"), every
// ds_read and MFMA in program order, counted lgkmcnt.  Row block i of a unit: read A(i+1) -> 2 MFMA -> read next unit's B(i)
// -> 6 MFMA; the A fragment is one block ahead (128 MFMA cycles), the B fragments one whole unit ahead.
template <int I, int SET, int FLAGS>
__device__ __forceinline__ void g4a_block(f32x4 (&acc)[8][8], bf16x8 (&fa)[2], bf16x8 (&fb)[2][8], int la_cur, int la_nxt, int lb_nxt) {
    if (!(FLAGS & 8)) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
    if constexpr (I < 7) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[(I + 1) & 1]) : "v"(la_cur), "n"((I + 1) * 1024));
    else asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0]) : "v"(la_nxt));
#define G4A_MF(J) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(fb[SET][J]), "v"(fa[I & 1]))
    G4A_MF(0); G4A_MF(1);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET ^ 1][I]) : "v"(lb_nxt), "n"(I * 1024));
    G4A_MF(2); G4A_MF(3); G4A_MF(4); G4A_MF(5); G4A_MF(6); G4A_MF(7);
#undef G4A_MF
}

// FLAGS: 1 one global->LDS load per row block instead of 8 at the top of the unit; ablations: 2 no loads, 4 no barrier, 8 no lgkmcnt,
// 16 (with 1) the loads go to a scratch VGPR instead of LDS (same global traffic, no LDS-DMA writes); 32 (with 17) each landed
// register is then stored with ds_write_b128 before it is reloaded (timing model of register-staged loads; results are wrong)
template <int FLAGS>
__global__ __launch_bounds__(256, 1) void g4a(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    const int nu = K / 32;
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long sw = (long)(((lane & 3) ^ swz64(lane >> 2)) << 3);
    const bf16* gA = A + (long)(m0 + wave * 64 + (lane >> 2)) * K + sw;
    const bf16* gB = B + (long)(n0 + wave * 64 + (lane >> 2)) * K + sw;
    auto stage = [&](int u, int slot) {
        char* dst = smem + slot * 32768 + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(gA + (long)q * 16 * K + (long)u * 32, dst + q * 1024);
            glds16(gB + (long)q * 16 * K + (long)u * 32, dst + 16384 + q * 1024);
        }
    };
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int kc = ((fq ^ swz64(fr)) << 4);
    const int la = lds0 + (wr * 128 + fr) * 64 + kc, lb = lds0 + 16384 + (wc * 128 + fr) * 64 + kc;
    bf16x8 fa[2], fb[2][8];
    stage(0, 0); stage(1, 1); stage(2, 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("ds_read_b128 %0, %1" : "=v"(fb[0][0]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(fb[0][1]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(fb[0][2]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(fb[0][3]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fb[0][4]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(fb[0][5]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(fb[0][6]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(fb[0][7]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0]) : "v"(la));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int lw = lds0 + wave * 4096 + lane * 16;              // FLAGS 32: where this lane's staged 16 bytes go (lane-linear, as the DMA writes)
    f32x4 junk[8];                                              // FLAGS 16: landing registers, live ("+v") until the final vmcnt(0)
#pragma unroll
    for (int i = 0; i < 8; ++i) junk[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto unit = [&](int u, auto setc) {
        constexpr int SET = decltype(setc)::value;
        const int us = u + 3 < nu ? u + 3 : nu - 1;             // past the end: unit nu-1 re-staged into a slot nobody reads
        if (!(FLAGS & 3)) stage(us, (u + 3) & 3);
        const bf16* sA = gA + (long)us * 32; const bf16* sB = gB + (long)us * 32;
        char* dst = smem + ((u + 3) & 3) * 32768 + wave * 4096;
        const int la_cur = la + (u & 3) * 32768, la_nxt = la + ((u + 1) & 3) * 32768, lb_nxt = lb + ((u + 1) & 3) * 32768;
#define G4A_BLK(I) do { if ((FLAGS & 19) == 1) { if ((I) < 4) glds16(sA + (long)(I) * 16 * K, dst + (I) * 1024); else glds16(sB + (long)((I) - 4) * 16 * K, dst + 16384 + ((I) - 4) * 1024); } \
                        if ((FLAGS & 19) == 17) { const bf16* gp = (I) < 4 ? sA + (long)(I) * 16 * K : sB + (long)((I) - 4) * 16 * K; \
                                                  if (FLAGS & 32) { asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); \
                                                      asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(lw + ((u + 3) & 3) * 32768), "v"(junk[I]), "n"(((I) & 3) * 1024 + ((I) >> 2) * 16384)); } \
                                                  asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(junk[I]) : "v"(gp)); } \
                        g4a_block<I, SET, FLAGS>(acc, fa, fb, la_cur, la_nxt, lb_nxt); } while (0)
        G4A_BLK(0); G4A_BLK(1); G4A_BLK(2); G4A_BLK(3); G4A_BLK(4); G4A_BLK(5); G4A_BLK(6); G4A_BLK(7);
#undef G4A_BLK
        if (FLAGS & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        if (!(FLAGS & 4)) __builtin_amdgcn_s_barrier();
    };
    for (int u = 0; u < nu; u += 2) { unit(u, IC<0>{}); unit(u + 1, IC<1>{}); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (FLAGS & 16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(junk[i]));
    }
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = clock64() - ck0; g_clk[1] = wall_clock64() - wk0; }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bf16* cp = C + (long)(m0 + wr * 128 + i * 16 + fr) * N + n0 + wc * 128 + j * 16 + fq * 4;
            const f32x4 v = acc[i][j];
            *(bf16x4*)cp = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        }
}

// g4a with a ring of RING slots (RING-2 units in flight across every barrier) and the A fragment two row blocks ahead (4 buffers).
template <int I, int SET>
__device__ __forceinline__ void g4b_block(f32x4 (&acc)[8][8], bf16x8 (&fa)[4], bf16x8 (&fb)[2][8], int la_cur, int la_nxt, int lb_nxt) {
    asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
    if constexpr (I < 6) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[(I + 2) & 3]) : "v"(la_cur), "n"((I + 2) * 1024));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[(I + 2) & 3]) : "v"(la_nxt), "n"((I - 6) * 1024));
#define G4B_MF(J) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(fb[SET][J]), "v"(fa[I & 3]))
    G4B_MF(0); G4B_MF(1);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET ^ 1][I]) : "v"(lb_nxt), "n"(I * 1024));
    G4B_MF(2); G4B_MF(3); G4B_MF(4); G4B_MF(5); G4B_MF(6); G4B_MF(7);
#undef G4B_MF
}

template <int RING>
__global__ __launch_bounds__(256, 1) void g4b(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    const int nu = K / 32;
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long sw = (long)(((lane & 3) ^ swz64(lane >> 2)) << 3);
    const bf16* gA = A + (long)(m0 + wave * 64 + (lane >> 2)) * K + sw;
    const bf16* gB = B + (long)(n0 + wave * 64 + (lane >> 2)) * K + sw;
    auto stage = [&](int u, int slot) {
        char* dst = smem + slot * 32768 + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(gA + (long)q * 16 * K + (long)u * 32, dst + q * 1024);
            glds16(gB + (long)q * 16 * K + (long)u * 32, dst + 16384 + q * 1024);
        }
    };
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int kc = ((fq ^ swz64(fr)) << 4);
    const int la = lds0 + (wr * 128 + fr) * 64 + kc, lb = lds0 + 16384 + (wc * 128 + fr) * 64 + kc;
    bf16x8 fa[4], fb[2][8];
#pragma unroll
    for (int u = 0; u < RING - 1; ++u) stage(u, u);
    if (RING == 5) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("ds_read_b128 %0, %1" : "=v"(fb[0][0]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(fb[0][1]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(fb[0][2]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(fb[0][3]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fb[0][4]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(fb[0][5]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(fb[0][6]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(fb[0][7]) : "v"(lb));
    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(fa[1]) : "v"(la));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int s_cur = 0;                                               // slot of unit u
    auto unit = [&](int u, auto setc) {
        constexpr int SET = decltype(setc)::value;
        const int us = u + RING - 1 < nu ? u + RING - 1 : nu - 1;     // past the end: unit nu-1 re-staged into a slot nobody reads
        const int s_nxt = s_cur + 1 == RING ? 0 : s_cur + 1, s_new = s_cur == 0 ? RING - 1 : s_cur - 1;
        const bf16* sA = gA + (long)us * 32; const bf16* sB = gB + (long)us * 32;
        char* dst = smem + s_new * 32768 + wave * 4096;
        const int la_cur = la + s_cur * 32768, la_nxt = la + s_nxt * 32768, lb_nxt = lb + s_nxt * 32768;
#define G4B_BLK(I) do { if ((I) < 4) glds16(sA + (long)(I) * 16 * K, dst + (I) * 1024); else glds16(sB + (long)((I) - 4) * 16 * K, dst + 16384 + ((I) - 4) * 1024); \
                        g4b_block<I, SET>(acc, fa, fb, la_cur, la_nxt, lb_nxt); } while (0)
        G4B_BLK(0); G4B_BLK(1); G4B_BLK(2); G4B_BLK(3); G4B_BLK(4); G4B_BLK(5); G4B_BLK(6); G4B_BLK(7);
#undef G4B_BLK
        if (RING == 5) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        s_cur = s_nxt;
    };
    for (int u = 0; u < nu; u += 2) { unit(u, IC<0>{}); unit(u + 1, IC<1>{}); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = clock64() - ck0; g_clk[1] = wall_clock64() - wk0; }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bf16* cp = C + (long)(m0 + wr * 128 + i * 16 + fr) * N + n0 + wc * 128 + j * 16 + fq * 4;
            const f32x4 v = acc[i][j];
            *(bf16x4*)cp = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// 4 waves x 128x128, BK = 64 per iteration, BOTH k-halves' fragments of both operands held in registers (128 VGPRs), two 64-KiB
// LDS buffers refilled IN PLACE by global->LDS DMA as soon as every wave has read an operand's half of the current buffer
// (data for iteration t+2).  Three barriers and three full lgkmcnt waits per 128 MFMAs, every ds_read issued >= 10 MFMAs
// before its wait, no VALU in the loop (SGPR base + constant 32-bit VGPR offsets for the loads).
// 128-B LDS rows, 16-B chunk ^ (row & 7).  FLAGS: ablations 2 no loads, 4 no barriers, 8 no ds_reads; 16 MFMA order with srcA fixed
template <int N, int BUF, int FLAGS>
__device__ __forceinline__ void g4h_step(f32x4 (&acc)[8][8], bf16x8 (&FA)[2][8], bf16x8 (&FB)[2][8], const int (&la)[2][2], const int (&lb)[2][2],
                                         const unsigned (&vA)[8], const unsigned (&vB)[8], const bf16* pA, const bf16* pB, int m0A, int m0B) {
    constexpr int h = N >> 6, n = N & 63, I = (FLAGS & 16) ? (n & 7) : (n >> 3), J = (FLAGS & 16) ? (n >> 3) : (n & 7);   // 16: srcA fixed over 8 MFMAs instead of srcB
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(FB[h][J]), "v"(FA[h][I]));
#define G4H_RD(dst, addr, off) do { if (!(FLAGS & 8)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off)); } while (0)
#define G4H_LD(voff, base, m0v) do { if (!(FLAGS & 2)) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff), "s"(base) : "memory"); } while (0)
#define G4H_BAR() do { if (!(FLAGS & 4)) __builtin_amdgcn_s_barrier(); } while (0)
    if constexpr (h == 0) {
        if constexpr (n < 16 && (n & 1)) G4H_RD(FB[1][n >> 1], lb[BUF][1], (n >> 1) * 2048);
        if constexpr (n == 20) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (n == 21) G4H_BAR();                                              // every wave has read all of B(cur)
        if constexpr (n >= 22 && n < 38 && !(n & 1)) G4H_LD(vB[(n - 22) >> 1], pB, m0B + ((n - 22) >> 1) * 1024);
        if constexpr (n >= 23 && n < 39 && (n & 1)) G4H_RD(FA[1][(n - 23) >> 1], la[BUF][1], ((n - 23) >> 1) * 2048);
        if constexpr (n == 50) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (n == 51) G4H_BAR();                                              // ... and all of A(cur)
        if constexpr (n == 52 || n == 55 || n == 58 || n == 61) G4H_LD(vA[(n - 52) / 3], pA, m0A + ((n - 52) / 3) * 1024);
    } else {
        if constexpr (n == 0) G4H_LD(vA[4], pA, m0A + 4 * 1024);
        if constexpr (n == 26) { if (FLAGS & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); }
        if constexpr (n == 27) G4H_BAR();                                              // the other buffer (loads of the previous iteration) has landed
        if constexpr (n >= 28 && n < 36) G4H_RD(FB[0][n - 28], lb[BUF ^ 1][0], (n - 28) * 2048);
        if constexpr (n >= 37 && n < 53 && (n & 1)) G4H_RD(FA[0][(n - 37) >> 1], la[BUF ^ 1][0], ((n - 37) >> 1) * 2048);
        if constexpr (n == 32 || n == 40 || n == 48) G4H_LD(vA[5 + (n - 32) / 8], pA, m0A + (5 + (n - 32) / 8) * 1024);
        if constexpr (n == 62) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef G4H_RD
#undef G4H_LD
#undef G4H_BAR
}

template <int BUF, int FLAGS, int... Ns>
__device__ __forceinline__ void g4h_iter(std::integer_sequence<int, Ns...>, f32x4 (&acc)[8][8], bf16x8 (&FA)[2][8], bf16x8 (&FB)[2][8], const int (&la)[2][2],
                                         const int (&lb)[2][2], const unsigned (&vA)[8], const unsigned (&vB)[8], const bf16* pA, const bf16* pB, int m0A, int m0B) {
    (g4h_step<Ns, BUF, FLAGS>(acc, FA, FB, la, lb, vA, vB, pA, pB, m0A, m0B), ...);
}

template <int FLAGS>
__global__ __launch_bounds__(256, 1) void g4h(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    const int nt = K / 64;
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    int la[2][2], lb[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pc = ((4 * h + fq) ^ (fr & 7)) << 4;
            la[b][h] = lds0 + b * 65536 + (wr * 128 + fr) * 128 + pc;
            lb[b][h] = lds0 + b * 65536 + 32768 + (wc * 128 + fr) * 128 + pc;
        }
    unsigned vA[8], vB[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned r = wave * 64 + q * 8 + (lane >> 3), ch = (lane & 7) ^ (lane >> 3);
        vA[q] = (r * (unsigned)K + ch * 8) * 2; vB[q] = vA[q];
    }
    const bf16* tA = A + (long)m0 * K; const bf16* tB = B + (long)n0 * K;
    const int mw = lds0 + wave * 8192;                           // this wave's 64 rows of an operand region
    auto stage = [&](int kt, int buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(mw + buf * 65536 + q * 1024), "v"(vA[q]), "s"(tA + (long)kt * 64) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(mw + buf * 65536 + 32768 + q * 1024), "v"(vB[q]), "s"(tB + (long)kt * 64) : "memory");
        }
    };
    bf16x8 FA[2][8], FB[2][8];
    stage(0, 0); stage(1, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FB[0][j]) : "v"(lb[0][0]), "n"(0)); lb[0][0] += 2048;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FA[0][i]) : "v"(la[0][0]), "n"(0)); la[0][0] += 2048;
    }
    lb[0][0] -= 8 * 2048; la[0][0] -= 8 * 2048;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    using Seq = std::make_integer_sequence<int, 128>;
    for (int t = 0; t < nt; t += 2) {
        {
            const int kl = t + 2 < nt ? t + 2 : nt - 1;          // past the end: the last tile again, into a buffer nobody reads any more
            g4h_iter<0, FLAGS>(Seq{}, acc, FA, FB, la, lb, vA, vB, tA + (long)kl * 64, tB + (long)kl * 64, mw, mw + 32768);
        }
        {
            const int kl = t + 3 < nt ? t + 3 : nt - 1;
            g4h_iter<1, FLAGS>(Seq{}, acc, FA, FB, la, lb, vA, vB, tA + (long)kl * 64, tB + (long)kl * 64, mw + 65536, mw + 65536 + 32768);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = clock64() - ck0; g_clk[1] = wall_clock64() - wk0; }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bf16* cp = C + (long)(m0 + wr * 128 + i * 16 + fr) * N + n0 + wc * 128 + j * 16 + fq * 4;
            const f32x4 v = acc[i][j];
            *(bf16x4*)cp = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        }
}

// 16 waves, 64x64 per wave like the reference, but v_mfma_f32_32x32x16_bf16 (2x2 blocks): half the MFMA instructions and
// operand register reads per flop.  128-B rows, chunk ^ ((row >> 1) & 7).
__global__ __launch_bounds__(1024, 4) void g16m32(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, hi = lane >> 5;
    const int tiles_m = M / TM;
    const int m0 = (blockIdx.x % tiles_m) * TM, n0 = (blockIdx.x / tiles_m) * TN;
    const int nt = K / 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int rsub = lane >> 3;
    auto stage = [&](int t, int boff) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int g = wave * 2 + q, r = g * 8 + rsub;
            const long sw = (long)(((lane & 7) ^ ((r >> 1) & 7)) << 3);
            glds16(A + (long)(m0 + r) * K + (long)t * 64 + sw, smem + boff + g * 1024);
            glds16(B + (long)(n0 + r) * K + (long)t * 64 + sw, smem + boff + 32768 + g * 1024);
        }
    };
    const int fsw = (l31 >> 1) & 7;
    const char* la = smem + (wm * 64 + l31) * 128;
    const char* lb = smem + 32768 + (wn * 64 + l31) * 128;
    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int boff = (t & 1) << 16;
        if (t + 1 < nt) stage(t + 1, 65536 - boff);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ko = (((ks * 2 + hi) ^ fsw) << 4);
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { fa[i] = *(const bf16x8*)(la + boff + ko + i * 4096); fb[i] = *(const bf16x8*)(lb + boff + ko + i * 4096); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = clock64() - ck0; g_clk[1] = wall_clock64() - wk0; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16* cp = C + (long)(m0 + wm * 64 + i * 32 + l31) * N + n0 + wn * 64 + j * 32 + 8 * q + 4 * hi;
                *(bf16x4*)cp = (bf16x4){(bf16)acc[i][j][4 * q], (bf16)acc[i][j][4 * q + 1], (bf16)acc[i][j][4 * q + 2], (bf16)acc[i][j][4 * q + 3]};
            }
}

// plain reference: 16 waves, BK = 64, two stages, __syncthreads
__device__ __forceinline__ void stage8(const bf16* base, long ld, int row0, int k0, char* lds, int group, int lane) {
    const int rsub = lane >> 3;
    glds16(base + (long)(row0 + group * 8 + rsub) * ld + k0 + (((lane & 7) ^ rsub) << 3), lds + group * 1024);
}
// MAP 0: block b -> (b % tiles_m, b / tiles_m); MAP (GM<<8|GN): each XCD (b & 7) owns a contiguous range of tile ids laid out in
// GM x GN super-tiles, so the 32 tiles an XCD runs at once share GM row panels and GN column panels in its L2
template <int MAP>
__global__ __launch_bounds__(1024, 4) void gref(const bf16* A, const bf16* B, bf16* C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = M / TM;
    int tm_ = blockIdx.x % tiles_m, tn_ = blockIdx.x / tiles_m;
    if (MAP) {
        constexpr int GM = MAP >> 8, GN = MAP & 255;
        const int id = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        const int sid = id / (GM * GN), within = id % (GM * GN), sgm = tiles_m / GM;
        tm_ = (sid % sgm) * GM + within % GM; tn_ = (sid / sgm) * GN + within / GM;
    }
    const int m0 = tm_ * TM, n0 = tn_ * TN;
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    constexpr int ST = 512 * 128;
    const int nt = K / 64;
    auto stage_half = [&](int t, int buf, int h) {
        char* base = smem + buf * ST + h * 256 * 128;
        const bf16* src = h ? B : A;
        const int r0 = h ? n0 : m0;
        stage8(src, K, r0, t * 64, base, wave, lane);
        stage8(src, K, r0, t * 64, base, 16 + wave, lane);
    };
    stage_half(0, 0, 0); stage_half(0, 0, 1);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const char* a_lds = smem + cur * ST;
        const char* b_lds = a_lds + 256 * 128;
        if (t + 1 < nt) { stage_half(t + 1, cur ^ 1, 0); stage_half(t + 1, cur ^ 1, 1); }
        for (int ks = 0; ks < 2; ++ks) {
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
        }
        __syncthreads();
    }
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = clock64() - ck0; g_clk[1] = wall_clock64() - wk0; }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            bf16* cp = C + (long)(m0 + wm * 64 + i * 16 + fr) * N + n0 + wn * 64 + j * 16 + fq * 4;
            for (int e = 0; e < 4; ++e) cp[e] = (bf16)acc[i][j][e];
        }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <typename KernT> float timeit(KernT kern, int threads, const bf16* A, const bf16* B, bf16* C, int M, int N, int K, int reps, int lds = 128 * 1024) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int tiles = (M / TM) * (N / TN);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(threads), lds, 0, A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(threads), lds, 0, A, B, C, M, N, K);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    if (M % 256 || N % 256 || K % 64 || K < 128) { printf("need M,N %% 256 == 0, K %% 64 == 0, K >= 128\n"); return 1; }
    bf16 *A, *B, *C, *C2;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2));
    CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&C2, (size_t)M * N * 2));
    const size_t na = (size_t)M * K, nb = (size_t)N * K, nmax = na > nb ? na : nb;
    unsigned short* h = (unsigned short*)malloc(nmax * 2);
    srand(1); for (size_t i = 0; i < na; ++i) h[i] = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
    CK(hipMemcpy(A, h, na * 2, hipMemcpyHostToDevice));
    srand(2); for (size_t i = 0; i < nb; ++i) h[i] = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
    CK(hipMemcpy(B, h, nb * 2, hipMemcpyHostToDevice));
    printf("M=%d N=%d K=%d\n", M, N, K);
    const double fl = 2.0 * M * N * K;
    float ms = timeit(gref<0>, 1024, A, B, C2, M, N, K, 20);
    { long long hk[2]; CK(hipMemcpyFromSymbol(hk, HIP_SYMBOL(g_clk), 16));
      printf("%-44s %8.1f us  %8.1f TF/s   main loop %.1f us, %.0f MHz\n", "reference (16 waves, 2 stages, syncthreads)", ms * 1000, fl / ms / 1e9, hk[1] / 100.0, 100.0 * hk[0] / hk[1]); }

    if (argc > 4 && argv[4][0] == 'h') {
        if ((K / 64) % 2) { printf("g4h needs K %% 128 == 0\n"); return 1; }
        for (int v = 0; v < 6; ++v) {
            CK(hipMemset(C, 0, (size_t)M * N * 2));
            ms = v == 0 ? timeit(g4h<0>, 256, A, B, C, M, N, K, 20) : v == 1 ? timeit(g4h<16>, 256, A, B, C, M, N, K, 20) : v == 2 ? timeit(g4h<2>, 256, A, B, C, M, N, K, 20)
               : v == 3 ? timeit(g4h<4>, 256, A, B, C, M, N, K, 20) : v == 4 ? timeit(g4h<8>, 256, A, B, C, M, N, K, 20) : timeit(gref<0>, 1024, A, B, C, M, N, K, 20);
            unsigned short* hc = (unsigned short*)malloc((size_t)M * N * 2);
            unsigned short* hr = (unsigned short*)malloc((size_t)M * N * 2);
            CK(hipMemcpy(hc, C, (size_t)M * N * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr, C2, (size_t)M * N * 2, hipMemcpyDeviceToHost));
            size_t bad = 0; double maxd = 0;
            for (size_t i = 0; i < (size_t)M * N; ++i) {
                unsigned int x = (unsigned int)hc[i] << 16, y = (unsigned int)hr[i] << 16;
                float fx, fy; memcpy(&fx, &x, 4); memcpy(&fy, &y, 4);
                const double d = fabs((double)fx - fy);
                if (d > maxd) maxd = d;
                if (d > 0.02 * fabs(fy) + 0.5) ++bad;
            }
            long long hk[2]; CK(hipMemcpyFromSymbol(hk, HIP_SYMBOL(g_clk), 16));
            printf("%-44s %8.1f us  %8.1f TF/s   max|diff| %.3f  mismatches %zu  main loop %.1f us, %.0f MHz\n",
                   v == 0 ? "4 waves, whole-k-half frags, in-place refill" : v == 1 ? "  same, MFMA order with srcA fixed" : v == 2 ? "  ablation: no global loads" : v == 3 ? "  ablation: no barriers" : v == 4 ? "  ablation: no ds_reads" : "reference again",
                   ms * 1000, fl / ms / 1e9, maxd, bad, hk[1] / 100.0, hk[1] ? 100.0 * hk[0] / hk[1] : 0.0);
            fflush(stdout);
            free(hc); free(hr);
        }
        return 0;
    }
#define RUNM(MAPV, name) ms = timeit(gref<MAPV>, 1024, A, B, C, M, N, K, 20); { long long hk[2]; CK(hipMemcpyFromSymbol(hk, HIP_SYMBOL(g_clk), 16)); \
    printf("%-44s %8.1f us  %8.1f TF/s   main loop %.1f us, %.0f MHz\n", name, ms * 1000, fl / ms / 1e9, hk[1] / 100.0, 100.0 * hk[0] / hk[1]); }
    if ((M / 256) % 16 || (N / 256) % 16 || ((M / 256) * (N / 256)) % 8) { printf("tile grid not a multiple of 16x16: mapping variants skipped\n"); goto after_maps; }
    RUNM((4 << 8) | 8, "reference, XCD map 4x8 super-tiles")
    RUNM((8 << 8) | 4, "reference, XCD map 8x4 super-tiles")
    RUNM((16 << 8) | 2, "reference, XCD map 16x2 (product today)")
    RUNM((2 << 8) | 16, "reference, XCD map 2x16")
    RUNM(0, "reference again")
after_maps:
    if (argc > 4) return 0;
    for (int rep = 0; rep < 3; ++rep) {                       // repeated: a staging race shows as a run-to-run difference
        CK(hipMemset(C, 0, (size_t)M * N * 2));
        ms = timeit(gpp<1>, 512, A, B, C, M, N, K, 20);
        unsigned short* hc = (unsigned short*)malloc((size_t)M * N * 2);
        unsigned short* hr = (unsigned short*)malloc((size_t)M * N * 2);
        CK(hipMemcpy(hc, C, (size_t)M * N * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr, C2, (size_t)M * N * 2, hipMemcpyDeviceToHost));
        size_t bad = 0; double maxd = 0;
        for (size_t i = 0; i < (size_t)M * N; ++i) {
            unsigned int x = (unsigned int)hc[i] << 16, y = (unsigned int)hr[i] << 16;
            float fx, fy; memcpy(&fx, &x, 4); memcpy(&fy, &y, 4);
            const double d = fabs((double)fx - fy);
            if (d > maxd) maxd = d;
            if (d > 0.02 * fabs(fy) + 0.5) ++bad;
        }
        printf("%-44s %8.1f us  %8.1f TF/s   max|diff| %.3f  mismatches %zu\n", "ping-pong 8 waves (setprio)", ms * 1000, fl / ms / 1e9, maxd, bad);
        free(hc); free(hr);
    }
#define RUNV(F, name) ms = timeit(gpp<F>, 512, A, B, C, M, N, K, 20); { long long hk[2]; CK(hipMemcpyFromSymbol(hk, HIP_SYMBOL(g_clk), 16)); \
    printf("%-44s %8.1f us  %8.1f TF/s   main loop %.1f us, %.0f MHz\n", name, ms * 1000, fl / ms / 1e9, hk[1] / 100.0, hk[1] ? 100.0 * hk[0] / hk[1] : 0.0); }
    for (int v = 0; v < 12; ++v) {
        CK(hipMemset(C, 0, (size_t)M * N * 2));
        ms = v == 11 ? timeit(g4b<5>, 256, A, B, C, M, N, K, 20, 160 * 1024) : v == 10 ? timeit(g4b<4>, 256, A, B, C, M, N, K, 20) : v == 9 ? timeit(g4a<0>, 256, A, B, C, M, N, K, 20) : v == 8 ? timeit(g4t<1>, 256, A, B, C, M, N, K, 20) : v == 7 ? timeit(g4t<3>, 256, A, B, C, M, N, K, 20) : v == 6 ? timeit(g4s<1>, 256, A, B, C, M, N, K, 20) : v == 5 ? timeit(g4s<0>, 256, A, B, C, M, N, K, 20) : v == 4 ? timeit(g16m32, 1024, A, B, C, M, N, K, 20) : v == 3 ? timeit(g4r<1>, 256, A, B, C, M, N, K, 20) : v == 2 ? timeit(g4r<0>, 256, A, B, C, M, N, K, 20) : v ? timeit(g4w<1>, 256, A, B, C, M, N, K, 20) : timeit(g4w<0>, 256, A, B, C, M, N, K, 20);
        unsigned short* hc = (unsigned short*)malloc((size_t)M * N * 2);
        unsigned short* hr = (unsigned short*)malloc((size_t)M * N * 2);
        CK(hipMemcpy(hc, C, (size_t)M * N * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr, C2, (size_t)M * N * 2, hipMemcpyDeviceToHost));
        size_t bad = 0; double maxd = 0;
        for (size_t i = 0; i < (size_t)M * N; ++i) {
            unsigned int x = (unsigned int)hc[i] << 16, y = (unsigned int)hr[i] << 16;
            float fx, fy; memcpy(&fx, &x, 4); memcpy(&fy, &y, 4);
            const double d = fabs((double)fx - fy);
            if (d > maxd) maxd = d;
            if (d > 0.02 * fabs(fy) + 0.5) ++bad;
        }
        long long hk[2]; CK(hipMemcpyFromSymbol(hk, HIP_SYMBOL(g_clk), 16));
        printf("%-44s %8.1f us  %8.1f TF/s   max|diff| %.3f  mismatches %zu  main loop %.1f us, %.0f MHz\n", v == 11 ? "asm K loop, ring 5, A two blocks ahead" : v == 10 ? "asm K loop, ring 4, A two blocks ahead" : v == 9 ? "4 waves, ring, asm K loop (AGPR acc)" : v == 8 ? "4 waves, ring, 16x16x32, streamed frags, pinned" : v == 7 ? "4 waves, ring, 16x16x32, streamed, pinned, branch-free" : v == 6 ? "4 waves, ring, 16x16x32, pinned" : v == 5 ? "4 waves, ring, 16x16x32" : v == 4 ? "16 waves 64x64, 32x32x16 MFMA" : v == 3 ? "4 waves, ring of 4 x BK32, pinned" : v == 2 ? "4 waves, ring of 4 x BK32" : v ? "4 waves 128x128, 32x32x16, frag prefetch" : "4 waves 128x128, 32x32x16", ms * 1000, fl / ms / 1e9, maxd, bad, hk[1] / 100.0, hk[1] ? 100.0 * hk[0] / hk[1] : 0.0);
        free(hc); free(hr);
    }
#define RUNA(F, name) ms = timeit(g4a<F>, 256, A, B, C, M, N, K, 20); { long long hk[2]; CK(hipMemcpyFromSymbol(hk, HIP_SYMBOL(g_clk), 16)); \
    printf("%-44s %8.1f us  %8.1f TF/s   main loop %.1f us, %.0f MHz\n", name, ms * 1000, fl / ms / 1e9, hk[1] / 100.0, hk[1] ? 100.0 * hk[0] / hk[1] : 0.0); }
    RUNA(1, "asm K loop, loads spread over row blocks")
    RUNA(2, "  ablation: no global loads in loop")
    RUNA(1 + 4, "  ablation: no barrier (spread loads)")
    RUNA(1 + 8, "  ablation: no lgkmcnt waits (spread loads)")
    RUNA(2 + 4 + 8, "  ablation: mfma + ds_read issue only")
    RUNA(1 + 16, "  ablation: loads land in VGPRs, not LDS")
    RUNA(1 + 16 + 32, "  timing model: VGPR landing + ds_write_b128 (2 units later)")
    if (argc > 5) return 0;
    RUNV(0, "ping-pong, no setprio")
    RUNV(1 + 2, "  no global loads in loop")
    RUNV(1 + 4, "  no ds_reads in loop")
    RUNV(1 + 2 + 4, "  no loads, no ds_reads (mfma + barriers)")
    RUNV(1 + 16, "  no mfma")
    RUNV(1 + 8, "  no stagger")
    RUNV(1 + 2 + 4 + 8, "  no stagger, no loads, no ds_reads")
    RUNV(1 + 2 + 4 + 8 + 32, "  mfma only (no barriers either)")
    return 0;
}
