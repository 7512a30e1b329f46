// Persistent 256x256 block-scaled fp8 GEMM for gfx950 (formats: fp8.hip).  C[M,N] (bf16) = act(A.B^T + bias) + R.
//
// Same skeleton as the bf16 persistent kernel (gemm.hip gemm_bf16_wp_kernel): one workgroup per CU walks tiles vid, vid+G, ...; 4 waves,
// one per SIMD, each owns a 128x128 quadrant (256 accumulator AGPRs); operands travel global -> LDS by global_load_lds_dwordx4 into two
// 64-KiB stage buffers (256 A rows + 256 B rows of 128 BYTES = one K-step of 128 fp8 elements: byte for byte the bf16 kernel's K-step,
// at twice the K); the pipeline never drains between tiles; the epilogue stores 16-byte bf16 chunks straight from the accumulators
// (column-interleaved weight fragments, gemm.hip WP_BOFF).
//
// What differs: one v_mfma_scale_f32_16x16x128_f8f6f4 consumes a whole 128-element row slice of each operand (8 VGPRs = the 16-byte chunks
// fq and 4 + fq of the lane's row, two ds_read_b128 -- the same two chunks the bf16 kernel reads as its k-halves), so a K-step is 64 MFMAs of 32 cycles instead of 128 of 16, and a fragment cannot be
// double-buffered by k-halves.  Fragments are instead reloaded IN PLACE as they die: MFMA order = weight columns 0..5 column-major, then
// row-major over columns 6,7; weight fragment J (< 6) is reloaded from the OTHER stage buffer (next K-step) right after its column,
// activation fragment I right after its two last products; the six reads that cannot be placed (FA[7], FB[6], FB[7]) trail into the next
// K-step, whose first MFMAs use fragments loaded long before (a counted lgkmcnt(6) wait at its first MFMA, a full one at its sixth).
// One barrier per K-step (after the wave's own DMA of the next K-step has landed): it certifies both "next buffer complete" and
// "everyone has finished reading the current buffer", after which the DMA for K-step t+2 overwrites the current buffer.
// Block scales: one dword per lane, K-step and 64-row group holds the E8M0 bytes of four MFMA tiles (OPSEL picks the byte); they ride
// two K-steps ahead in registers like the DMA.
// LDS swizzle: exactly the bf16 kernel's (activation rows: chunk ^ (row & 7); weight rows: chunk ^ ((row & 3) | (row >> 3 & 1) << 2)).
#include "common.h"
#include "avllm_internal.h"
#include "gemm_shared.h"
#include <stdlib.h>
#include <utility>
#include <type_traits>

namespace {

using avg::epilogue_fast8; using avg::xcd_remap; using avg::tile_coords;
__device__ __forceinline__ int f8_groups_dev(int R) { return (R + 255) / 256 * 4; }

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) int v4i;

constexpr int TM = 256, TN = 256, KB = 128;            // tile, bytes (= fp8 elements) per row and K-step
constexpr int STAGE = (TM + TN) * KB;                  // 64 KiB
constexpr int F8_LDS = 2 * STAGE;

struct F8Args {
    const uint8_t* A; const uint8_t* B; const uint32_t* SA; const uint32_t* SB;
    long lda, ldb;
    int K, RBA, RBB, dbg;
    void* C; long ldc; const void* bias; const void* R; long ldr; int act; int M, N;
    uint8_t* Cq; uint32_t* SCq; long ldcq;                             // quantised output (see include/avllm.h)
};

#define F8_BOFF(j) ((((j) >> 1) * 32 + ((j) & 1) * 4) * 128)
#define F8_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define F8_LD(voff, base, m0v) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff), "s"(base) : "memory")
#define F8_LDS32(dst, voff, base) asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory")

// opsel of the weight-side scale (first MFMA operand) = J & 3, of the activation-side scale = I & 3
template <int OA, int OB>
__device__ __forceinline__ void f8_mfma(f32x4& acc, const v8i fb, const v8i fa, int sb, int sa) {
#define F8_MFMA_ASM(SEL)                                                                                                                          \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 " SEL : "+a"(acc) : "v"(fb), "v"(fa), "v"(sb), "v"(sa))
    // op_sel:[a0,b0,0] op_sel_hi:[a1,b1,0]: operand a's byte = a1 * 2 + a0
    if constexpr (OA == 0 && OB == 0) { F8_MFMA_ASM("op_sel_hi:[0,0,0]"); }
    else if constexpr (OA == 1 && OB == 0) { F8_MFMA_ASM("op_sel:[1,0,0] op_sel_hi:[0,0,0]"); }
    else if constexpr (OA == 2 && OB == 0) { F8_MFMA_ASM("op_sel:[0,0,0] op_sel_hi:[1,0,0]"); }
    else if constexpr (OA == 3 && OB == 0) { F8_MFMA_ASM("op_sel:[1,0,0] op_sel_hi:[1,0,0]"); }
    else if constexpr (OA == 0 && OB == 1) { F8_MFMA_ASM("op_sel:[0,1,0] op_sel_hi:[0,0,0]"); }
    else if constexpr (OA == 1 && OB == 1) { F8_MFMA_ASM("op_sel:[1,1,0] op_sel_hi:[0,0,0]"); }
    else if constexpr (OA == 2 && OB == 1) { F8_MFMA_ASM("op_sel:[0,1,0] op_sel_hi:[1,0,0]"); }
    else if constexpr (OA == 3 && OB == 1) { F8_MFMA_ASM("op_sel:[1,1,0] op_sel_hi:[1,0,0]"); }
    else if constexpr (OA == 0 && OB == 2) { F8_MFMA_ASM("op_sel:[0,0,0] op_sel_hi:[0,1,0]"); }
    else if constexpr (OA == 1 && OB == 2) { F8_MFMA_ASM("op_sel:[1,0,0] op_sel_hi:[0,1,0]"); }
    else if constexpr (OA == 2 && OB == 2) { F8_MFMA_ASM("op_sel:[0,0,0] op_sel_hi:[1,1,0]"); }
    else if constexpr (OA == 3 && OB == 2) { F8_MFMA_ASM("op_sel:[1,0,0] op_sel_hi:[1,1,0]"); }
    else if constexpr (OA == 0 && OB == 3) { F8_MFMA_ASM("op_sel:[0,1,0] op_sel_hi:[0,1,0]"); }
    else if constexpr (OA == 1 && OB == 3) { F8_MFMA_ASM("op_sel:[1,1,0] op_sel_hi:[0,1,0]"); }
    else if constexpr (OA == 2 && OB == 3) { F8_MFMA_ASM("op_sel:[0,1,0] op_sel_hi:[1,1,0]"); }
    else { F8_MFMA_ASM("op_sel:[1,1,0] op_sel_hi:[1,1,0]"); }
#undef F8_MFMA_ASM
}

struct F8Frag { v4i lo, hi; };

template <int I, int N, class F> __device__ __forceinline__ void f8_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); f8_static_for<I + 1, N>(f); }
}

// Slot N of a K-step.  ONE instantiation serves every K-step of every tile (the tile boundary is a cold block inside the K-step loop: a
// separate copy of this 64-MFMA body for "first K-step of a tile" made the register allocator permute the 128 fragment registers through
// scratch at every tile boundary).  `relaxed`: this is the first K-step after a full epilogue, whose stores may stay in flight (gemm.hip
// wpgemm_step MODE 2).  ad[] = this lane's fragment addresses in the OTHER buffer (next K-step): A lo / A hi /
// B lo / B hi (the two 16-byte halves of a lane's 32 bytes are swizzled apart, hence two bases per operand).
template <int N>
__device__ __forceinline__ void f8_step(f32x4 (&acc)[8][8], F8Frag (&FA)[8], F8Frag (&FB)[8], int (&sCur)[4], int (&sNext)[4],
                                        int (&sN2)[4], const int (&ad)[4], const unsigned (&vA)[8], const unsigned (&vB)[8], const uint8_t* pA,
                                        const uint8_t* pB, int m0A, int m0B, const unsigned (&vS)[4], const uint32_t* pSA, const uint32_t* pSB, bool relaxed) {
    constexpr int I = N < 48 ? (N & 7) : ((N - 48) >> 1), J = N < 48 ? (N >> 3) : 6 + ((N - 48) & 1);
    // the first MFMAs use fragments read long before the six reads that trailed out of the previous K-step (FA[7], FB[6], FB[7])
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    f8_mfma<(J & 3), (I & 3)>(acc[I][J], __builtin_shufflevector(FB[J].lo, FB[J].hi, 0, 1, 2, 3, 4, 5, 6, 7),
                                           __builtin_shufflevector(FA[I].lo, FA[I].hi, 0, 1, 2, 3, 4, 5, 6, 7), sCur[2 + (J >> 2)], sCur[I >> 2]);
    // ---- memory side
    if constexpr (N == 5) {
        // the wave's own DMA (and scale words) of the next K-step have landed, and its trailing reads of the CURRENT buffer are done: after
        // the barrier below the current buffer may be overwritten
        if (relaxed) asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) sNext[k] = sN2[k];
    }
    if constexpr (N == 6) __builtin_amdgcn_s_barrier();       // next buffer complete for every wave; current buffer read by every wave
    if constexpr (N >= 8 && N < 16 && !(N & 1)) {             // scale words of K-step t+2: SA group 0/1, SB group 0/1
        constexpr int k = (N - 8) >> 1;
        if constexpr (k < 2) F8_LDS32(sN2[k], vS[k], pSA); else F8_LDS32(sN2[k], vS[k], pSB);
    }
    if constexpr (N >= 16 && N < 48 && !(N & 1)) {            // DMA of K-step t+2 into the CURRENT buffer: 8 A loads, then 8 B loads
        constexpr int k = (N - 16) >> 1;
        if constexpr (k < 8) F8_LD(vA[k], pA, m0A + k * 1024); else F8_LD(vB[k - 8], pB, m0B + (k - 8) * 1024);
    }
    // ---- fragment reloads for the next K-step, in place: weight fragment j after its column (slots 8j+9, 8j+10), activation fragment i after
    // its two last products (slots 50+2i, 51+2i)
    if constexpr (N >= 9 && N < 50 && ((N - 9) & 7) == 0) { constexpr int j = (N - 9) >> 3; F8_RD(FB[j].lo, ad[2], F8_BOFF(j)); }
    if constexpr (N >= 10 && N < 51 && ((N - 10) & 7) == 0) { constexpr int j = (N - 10) >> 3; F8_RD(FB[j].hi, ad[3], F8_BOFF(j)); }
    if constexpr (N >= 50 && !(N & 1)) { constexpr int i = (N - 50) >> 1; F8_RD(FA[i].lo, ad[0], i * 2048); }
    if constexpr (N >= 51 && (N & 1)) { constexpr int i = (N - 51) >> 1; F8_RD(FA[i].hi, ad[1], i * 2048); }
}

template <int... Ns>
__device__ __forceinline__ void f8_kstep(std::integer_sequence<int, Ns...>, f32x4 (&acc)[8][8], F8Frag (&FA)[8], F8Frag (&FB)[8],
                                         int (&sCur)[4], int (&sNext)[4], int (&sN2)[4], const int (&ad)[4], const unsigned (&vA)[8], const unsigned (&vB)[8],
                                         const uint8_t* pA, const uint8_t* pB, int m0A, int m0B, const unsigned (&vS)[4], const uint32_t* pSA,
                                         const uint32_t* pSB, bool relaxed) {
    (f8_step<Ns>(acc, FA, FB, sCur, sNext, sN2, ad, vA, vB, pA, pB, m0A, m0B, vS, pSA, pSB, relaxed), ...);
    // the reads that could not be placed: FA[7] (died at slot 63), FB[6] (62), FB[7] (63)
    F8_RD(FA[7].lo, ad[0], 7 * 2048); F8_RD(FA[7].hi, ad[1], 7 * 2048);
    F8_RD(FB[6].lo, ad[2], F8_BOFF(6)); F8_RD(FB[6].hi, ad[3], F8_BOFF(6));
    F8_RD(FB[7].lo, ad[2], F8_BOFF(7)); F8_RD(FB[7].hi, ad[3], F8_BOFF(7));
#pragma unroll
    for (int k = 0; k < 4; ++k) sCur[k] = sNext[k];
    asm volatile("s_nop 1" ::: "memory");
}

__global__ __launch_bounds__(256, 1) void gemm_f8_wp_kernel(F8Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
    const int ntiles = tiles_m * tiles_n, G = gridDim.x;             // G <= ntiles (dispatcher)
    const int nt = g.K / KB;                                          // >= 2 (dispatcher)
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    // fragment addresses: the lane's 32 operand bytes are the 16-byte chunks fq and 4 + fq of its row (fp8.hip "Operand layout"), each ^ key(row)
    const int keyA = fr & 7;
    const int keyB = (fr & 3) | (((fr >> 2) & 1) << 2);
    const int rowA = lds0 + (wr * 128 + fr) * 128, rowB = lds0 + TM * 128 + (wc * 128 + 8 * (fr >> 2) + (fr & 3)) * 128;
    const int base[4] = {rowA + ((fq ^ keyA) << 4), rowA + (((4 + fq) ^ keyA) << 4), rowB + ((fq ^ keyB) << 4), rowB + (((4 + fq) ^ keyB) << 4)};
    int ad[4] = {base[0] + STAGE, base[1] + STAGE, base[2] + STAGE, base[3] + STAGE};      // current buffer = 0: "other" = buffer 1
    int bo = 0;                                                      // byte offset of the current buffer
    const int mw = lds0 + wave * 8192;
    const unsigned l3 = lane >> 3, r0 = wave * 64 + l3;
    const unsigned chA = ((lane & 7) ^ l3) << 4;                                              // activation rows: key = row & 7
    const unsigned chB0 = ((lane & 7) ^ (l3 & 3)) << 4, chB1 = ((lane & 7) ^ ((l3 & 3) | 4)) << 4;      // weight row wave*64 + 8q + l3: key = (l3 & 3) | (q & 1) << 2

    unsigned vA1[8], vB1[8], vS[4];
    const uint8_t *tA1, *tB1;
    int lvid = blockIdx.x, lt = 0, lm0 = 0, ln0 = 0;
    auto set_ctx = [&](int vid) __attribute__((always_inline)) {
        int tm, tn;
        tile_coords(xcd_remap(vid, ntiles), tiles_m, tiles_n, tm, tn);
        lm0 = tm * TM; ln0 = tn * TN;
        const unsigned ma = g.M - 1 - lm0, mb = g.N - 1 - ln0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned r = r0 + q * 8;
            vA1[q] = __umul24(r < ma ? r : ma, (unsigned)g.lda) + chA;
            vB1[q] = __umul24(r < mb ? r : mb, (unsigned)g.ldb) + ((q & 1) ? chB1 : chB0);
        }
        tA1 = g.A + (long)lm0 * g.lda; tB1 = g.B + (long)ln0 * g.ldb;
        // scale words of this tile: byte offset of (group, fq, fr) inside a K-step's plane; group = tile's first + 2 * (wave's half) + {0, 1}
        vS[0] = ((((lm0 >> 6) + wr * 2 + 0) * 4 + fq) * 16 + fr) * 4; vS[1] = ((((lm0 >> 6) + wr * 2 + 1) * 4 + fq) * 16 + fr) * 4;
        vS[2] = ((((ln0 >> 6) + wc * 2 + 0) * 4 + fq) * 16 + fr) * 4; vS[3] = ((((ln0 >> 6) + wc * 2 + 1) * 4 + fq) * 16 + fr) * 4;
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if (++lt < nt) return;
        if (lvid + G < ntiles) { lvid += G; set_ctx(lvid); lt = 0; }
        else lt = nt - 1;                                            // nothing left: keep re-fetching the last K-step (never read)
    };
    const long planeA = (long)g.RBA * 256, planeB = (long)g.RBB * 256;      // bytes of one K-step's scale plane (groups x 4 x 16 words)
    int sCur[4], sNext[4] = {0, 0, 0, 0}, sN2[4];
    set_ctx(lvid);
    {   // de-phase the workgroups (see gemm_bf16_wp_kernel)
        const int phases = (g.dbg >> 4) ? (g.dbg >> 4) : (ntiles >= 6 * G ? 2 : 1);
        if (phases > 1) {
            const int n = ((blockIdx.x >> 3) % phases) * (nt * 2100 + 11000) / (phases * 6400);
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(100);
        }
    }
    // prologue: K-steps 0 and 1 of the first tile into buffers 0 and 1, their scale words into sCur / sN2
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint8_t *pa = tA1 + (long)lt * KB, *pb = tB1 + (long)lt * KB;
        const uint32_t* psa = (const uint32_t*)((const char*)g.SA + lt * planeA);
        const uint32_t* psb = (const uint32_t*)((const char*)g.SB + lt * planeB);
        if (s == 0) { F8_LDS32(sCur[0], vS[0], psa); F8_LDS32(sCur[1], vS[1], psa); F8_LDS32(sCur[2], vS[2], psb); F8_LDS32(sCur[3], vS[3], psb); }
        else { F8_LDS32(sN2[0], vS[0], psa); F8_LDS32(sN2[1], vS[1], psa); F8_LDS32(sN2[2], vS[2], psb); F8_LDS32(sN2[3], vS[3], psb); }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            F8_LD(vA1[q], pa, mw + s * STAGE + q * 1024);
            F8_LD(vB1[q], pb, mw + s * STAGE + TM * 128 + q * 1024);
        }
        advance();
    }
    f32x4 acc[8][8];
    F8Frag FA[8], FB[8];                                             // a lane's 32 operand bytes as the two 16-byte halves it reads
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");               // K-step 0 (and its scale words) have landed; K-step 1's 20 loads may be in flight
    __builtin_amdgcn_s_barrier();
#define F8_PRO_A(I) F8_RD(FA[I].lo, base[0], (I) * 2048); F8_RD(FA[I].hi, base[1], (I) * 2048)
#define F8_PRO_B(J) F8_RD(FB[J].lo, base[2], F8_BOFF(J)); F8_RD(FB[J].hi, base[3], F8_BOFF(J))
    F8_PRO_B(0); F8_PRO_B(1); F8_PRO_B(2); F8_PRO_B(3); F8_PRO_B(4); F8_PRO_B(5); F8_PRO_B(6); F8_PRO_B(7);
    F8_PRO_A(0); F8_PRO_A(1); F8_PRO_A(2); F8_PRO_A(3); F8_PRO_A(4); F8_PRO_A(5); F8_PRO_A(6); F8_PRO_A(7);
#undef F8_PRO_A
#undef F8_PRO_B
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    using Seq = std::make_integer_sequence<int, 64>;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int my_tiles = (ntiles - 1 - (int)blockIdx.x) / G + 1;
    bool relaxed = false;                                            // the previous K-step was followed by a full epilogue (32 stores per wave)
    int vid = blockIdx.x, t = 0;
#pragma unroll 1
    for (int step = 0; step < my_tiles * nt; ++step) {
        {
            const uint32_t* psa = (const uint32_t*)((const char*)g.SA + lt * planeA);
            const uint32_t* psb = (const uint32_t*)((const char*)g.SB + lt * planeB);
            f8_kstep(Seq{}, acc, FA, FB, sCur, sNext, sN2, ad, vA1, vB1, tA1 + (long)lt * KB, tB1 + (long)lt * KB, mw + bo, mw + bo + TM * 128, vS, psa, psb,
                     relaxed);
            const int d = bo ? -STAGE : STAGE;                        // the buffers swap roles
#pragma unroll
            for (int k = 0; k < 4; ++k) ad[k] -= d;
            bo += d;
            advance();
            relaxed = false;
        }
        if (++t < nt) continue;
        t = 0;
        // ---- tile boundary.  Result latency of the last MFMAs (invisible to the compiler's hazard recogniser): nops, and every accumulator
        // named as an in/out operand so that compiler-generated readers stay behind them
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#define F8_PIN8(I) asm volatile("" : "+a"(acc[I][0]), "+a"(acc[I][1]), "+a"(acc[I][2]), "+a"(acc[I][3]), "+a"(acc[I][4]), "+a"(acc[I][5]), "+a"(acc[I][6]), "+a"(acc[I][7]))
        F8_PIN8(0); F8_PIN8(1); F8_PIN8(2); F8_PIN8(3); F8_PIN8(4); F8_PIN8(5); F8_PIN8(6); F8_PIN8(7);
        // epilogue straight from the accumulators (as gemm_bf16_wp_kernel): lane (fr, fq), row block i, column pair p -> 8 consecutive columns
        int tm, tn;
        tile_coords(xcd_remap(vid, ntiles), tiles_m, tiles_n, tm, tn);
        const int m0 = tm * TM + wr * 128 + fr, n = tn * TN + wc * 128 + fq * 8;
        relaxed = (tm + 1) * TM <= g.M && (tn + 1) * TN <= g.N && !(g.dbg & 3);
        float b[4][8];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int c = 0; c < 8; ++c) b[p][c] = 0.f;
            if (g.bias && n + 32 * p < g.N) load_f<8>((const bf16*)g.bias + n + 32 * p, b[p]);
        }
        bf16* const cp0 = (bf16*)g.C + (long)m0 * g.ldc + n;
        const bf16* const rp0 = g.R ? (const bf16*)g.R + (long)m0 * g.ldr + n : nullptr;
        const long ldc = g.ldc, ldr = g.R ? g.ldr : 0;
        auto run = [&](auto actc) __attribute__((always_inline)) {
            constexpr int ACT = decltype(actc)::value;
#pragma clang loop unroll(full)
            for (int i = 0; i < 8; ++i) {
                const bool mrow = m0 + i * 16 < g.M;
#pragma clang loop unroll(full)
                for (int p = 0; p < 4; ++p) {
                    // the accumulators are "redefined" here for the compiler: otherwise every epilogue variant's 256 accumulator reads are hoisted above
                    // the variant branch as common subexpressions (256 live registers: 57 spilled VGPRs, 460 scratch accesses around the epilogue)
                    asm volatile("" : "+a"(acc[i][2 * p]), "+a"(acc[i][2 * p + 1]));
                    if (mrow && n + 32 * p < g.N && !(g.dbg & 1))
                        epilogue_fast8<ACT>(acc[i][2 * p], acc[i][2 * p + 1], b[p], g.bias != nullptr, rp0 ? rp0 + (i * 16) * ldr + 32 * p : nullptr,
                                            cp0 + (i * 16) * ldc + 32 * p);
                }
            }
        };
        // Quantised output: act(acc + bias) -> e4m3 + one E8M0 per 32 columns, from the fp32 values (the separate quantiser pass over a bf16
        // copy costs a read of 2 and a write of 1 byte per element: 53 ms of a 541 ms config-5 step for the ViT-L fc1 outputs alone).
        // A 32-column block of a row lives in the 4 lanes fq = 0..3 of one fr: amax through two cross-row shuffles.  Scale image (layout 0):
        // word ((t * RB + rb) * 4 + blk) * 16 + fr holds the bytes of rows 64 rb + 16 i' + fr, i' = 0..3, for column block blk of K-step t of the
        // CONSUMER (128 columns): this lane's row blocks i = 4 a + i' fill whole words; lane fq writes the words of block p == fq.
        auto run_q = [&](auto actc) __attribute__((always_inline)) {
            constexpr int ACT = decltype(actc)::value;
            // the eight scale words of the lane as named scalars, updated through compile-time indices (f8_static_for): as an array indexed by
            // the unrolled loop variables they stayed in scratch memory -- a scratch load + vmcnt(0) + scratch store behind every chunk's
            // global store, i.e. one store acknowledgement of latency per chunk
            uint32_t sw00 = 0u, sw01 = 0u, sw02 = 0u, sw03 = 0u, sw10 = 0u, sw11 = 0u, sw12 = 0u, sw13 = 0u;
            uint8_t* const qp0 = g.Cq + (long)m0 * g.ldcq + n;
            const long ldcq = g.ldcq;
            f8_static_for<0, 8>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const bool mrow = m0 + i * 16 < g.M;
                f8_static_for<0, 4>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int p = decltype(pc)::value;
                    asm volatile("" : "+a"(acc[i][2 * p]), "+a"(acc[i][2 * p + 1]));
                    const f32x4 lo = acc[i][2 * p], hi = acc[i][2 * p + 1];
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (g.bias != nullptr) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] += b[p][c];
                    }
                    if constexpr (ACT != AV_ACT_NONE) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = act_apply_fast(v[c], ACT);
                    }
                    float amax = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) amax = fmaxf(amax, fabsf(v[c]));
                    amax = fmaxf(amax, __shfl_xor(amax, 16));
                    amax = fmaxf(amax, __shfl_xor(amax, 32));
                    int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;           // floor(log2 amax) - 8 (oracle/mxfp8.py, fp8.hip mx_quant_kernel)
                    e = e < -127 ? -127 : (e > 127 ? 127 : e);
                    const float invs = __uint_as_float((uint32_t)(127 - e) << 23);
                    float a8[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) a8[c] = fminf(fmaxf(v[c] * invs, -448.f), 448.f);
                    int r0 = 0, r1 = 0;
                    r0 = __builtin_amdgcn_cvt_pk_fp8_f32(a8[0], a8[1], r0, false);
                    r0 = __builtin_amdgcn_cvt_pk_fp8_f32(a8[2], a8[3], r0, true);
                    r1 = __builtin_amdgcn_cvt_pk_fp8_f32(a8[4], a8[5], r1, false);
                    r1 = __builtin_amdgcn_cvt_pk_fp8_f32(a8[6], a8[7], r1, true);
                    const bool on = mrow && n + 32 * p < g.N;
                    if (on) {
                        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                        *(u32x2*)(qp0 + (long)(i * 16) * ldcq + 32 * p) = (u32x2){(uint32_t)r0, (uint32_t)r1};
                    }
                    const uint32_t bits = on ? (uint32_t)(e + 127) << (8 * (i & 3)) : 0u;
                    if constexpr (i < 4) { if constexpr (p == 0) sw00 |= bits; else if constexpr (p == 1) sw01 |= bits; else if constexpr (p == 2) sw02 |= bits; else sw03 |= bits; }
                    else { if constexpr (p == 0) sw10 |= bits; else if constexpr (p == 1) sw11 |= bits; else if constexpr (p == 2) sw12 |= bits; else sw13 |= bits; }
                });
            });
            const int t = tn * (TN / 128) + wc, RBo = f8_groups_dev(g.M);
            const uint32_t wv0 = fq == 0 ? sw00 : fq == 1 ? sw01 : fq == 2 ? sw02 : sw03;      // selects on scalars: an array indexed by fq lives in scratch
            const uint32_t wv1 = fq == 0 ? sw10 : fq == 1 ? sw11 : fq == 2 ? sw12 : sw13;
            const bool col_ok = tn * TN + wc * 128 + 32 * fq < g.N;
            const int rb0 = tm * (TM / 64) + wr * 2;
            if (col_ok && rb0 < RBo) g.SCq[(((long)t * RBo + rb0) * 4 + fq) * 16 + fr] = wv0;
            if (col_ok && rb0 + 1 < RBo) g.SCq[(((long)t * RBo + rb0 + 1) * 4 + fq) * 16 + fr] = wv1;
        };
        // full tile, bf16 output: the lean forms of gemm_bf16_wp_kernel (no bounds test, no select, no branch per chunk; with a residual, all 32
        // residual chunks of the lane requested in one burst into registers that are dead here and consumed behind ONE constant wait count:
        // 31 - c younger loads + c stores behind chunk c's load)
        const bool full = relaxed;
        auto run_full = [&](auto actc, auto hasbc) __attribute__((always_inline)) {
            constexpr int ACT = decltype(actc)::value;
            constexpr bool HASB = decltype(hasbc)::value;
#pragma clang loop unroll(full)
            for (int i = 0; i < 8; ++i) {
                bf16* const cpi = cp0 + (long)(i * 16) * ldc;
#pragma clang loop unroll(full)
                for (int p = 0; p < 4; ++p) {
                    asm volatile("" : "+a"(acc[i][2 * p]), "+a"(acc[i][2 * p + 1]));
                    const f32x4 lo = acc[i][2 * p], hi = acc[i][2 * p + 1];
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if constexpr (HASB) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] += b[p][c];
                    }
                    if constexpr (ACT != AV_ACT_NONE) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = act_apply_fast(v[c], ACT);
                    }
                    store_f<8>(cpi + 32 * p, v);
                }
            }
        };
        auto run_res_full = [&](auto hasbc) __attribute__((always_inline)) {
            constexpr bool HASB = decltype(hasbc)::value;
            // one row block of residual chunks ahead of the stores (this kernel carries the next tile's 128 fragment registers through the
            // epilogue, so the whole-tile burst of the bf16 kernel does not fit): asm loads + hand-counted waits, see gemm.hip
            constexpr int RD = 1;
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
            u32x4_ rr[RD + 1][4];
            auto fetch = [&](int i) __attribute__((always_inline)) {
                const bf16* rp = rp0 + (long)(i * 16) * ldr;
                asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"
                             "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
                             : "=&v"(rr[i % (RD + 1)][0]), "=&v"(rr[i % (RD + 1)][1]), "=&v"(rr[i % (RD + 1)][2]), "=&v"(rr[i % (RD + 1)][3]) : "v"(rp) : "memory");
            };
            fetch(0);
#pragma clang loop unroll(full)
            for (int i = 0; i < 8; ++i) {
                if (i + RD < 8) fetch(i + RD);
                bf16* const cpi = cp0 + (long)(i * 16) * ldc;
                const int nw = 3 + 4 * (RD < 7 - i ? RD : 7 - i) + 4 * (RD < i ? RD : i);
#pragma clang loop unroll(full)
                for (int p = 0; p < 4; ++p) {
                    u32x4_& x = rr[i % (RD + 1)][p];
                    if (nw == 7) asm volatile("s_waitcnt vmcnt(7)" : "+v"(x) :: "memory");
                    else if (nw == 11) asm volatile("s_waitcnt vmcnt(11)" : "+v"(x) :: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(x) :: "memory");
                    const bf16x8 r = __builtin_bit_cast(bf16x8, x);
                    asm volatile("" : "+a"(acc[i][2 * p]), "+a"(acc[i][2 * p + 1]));
                    const f32x4 lo = acc[i][2 * p], hi = acc[i][2 * p + 1];
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if constexpr (HASB) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] += b[p][c];
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] += (float)r[c];
                    store_f<8>(cpi + 32 * p, v);
                }
            }
        };
        const std::true_type yes{};
        const std::false_type no{};
        const bool hb = g.bias != nullptr;
        if (g.Cq) {
            if (g.act == AV_ACT_NONE) run_q(std::integral_constant<int, AV_ACT_NONE>{});
            else if (g.act == AV_ACT_GELU) run_q(std::integral_constant<int, AV_ACT_GELU>{});
            else if (g.act == AV_ACT_QUICK_GELU) run_q(std::integral_constant<int, AV_ACT_QUICK_GELU>{});
            else run_q(std::integral_constant<int, AV_ACT_SILU>{});
        } else if (full && !g.R && g.act == AV_ACT_NONE && !hb) run_full(std::integral_constant<int, AV_ACT_NONE>{}, no);
        else if (full && !g.R && g.act == AV_ACT_NONE && hb) run_full(std::integral_constant<int, AV_ACT_NONE>{}, yes);
        else if (full && !g.R && g.act == AV_ACT_GELU && hb) run_full(std::integral_constant<int, AV_ACT_GELU>{}, yes);
        else if (full && !g.R && g.act == AV_ACT_QUICK_GELU && hb) run_full(std::integral_constant<int, AV_ACT_QUICK_GELU>{}, yes);
        else if (full && g.R && g.act == AV_ACT_NONE && hb) run_res_full(yes);
        else if (full && g.R && g.act == AV_ACT_NONE) run_res_full(no);
        else if (g.act == AV_ACT_NONE) run(std::integral_constant<int, AV_ACT_NONE>{});
        else if (g.act == AV_ACT_GELU) run(std::integral_constant<int, AV_ACT_GELU>{});
        else if (g.act == AV_ACT_QUICK_GELU) run(std::integral_constant<int, AV_ACT_QUICK_GELU>{});
        else run(std::integral_constant<int, AV_ACT_SILU>{});
        // The next K-step's operand fragments are read AGAIN here instead of being carried through the epilogue (the K loop loaded them in
        // place during the tile's last K-step; their LDS buffer stays untouched until that K-step runs: its own DMA refills it): 128 registers
        // the epilogue can use -- it spilled 26 - 57 of them otherwise, and every scratch reload between the tile's global stores is a
        // compiler-made vmcnt(0), i.e. a wait for the stores' acknowledgements.  Unconditional ("=v" outputs: the old values are dead).
        {
            const int dx = bo ? STAGE : -STAGE;                       // ad[] already points at the other buffer: undo the last swap
            const int a0 = ad[0] + dx, a1 = ad[1] + dx, b0 = ad[2] + dx, b1 = ad[3] + dx;
#define F8_RE_A(I) F8_RD(FA[I].lo, a0, (I) * 2048); F8_RD(FA[I].hi, a1, (I) * 2048)
#define F8_RE_B(J) F8_RD(FB[J].lo, b0, F8_BOFF(J)); F8_RD(FB[J].hi, b1, F8_BOFF(J))
            F8_RE_B(0); F8_RE_B(1); F8_RE_B(2); F8_RE_B(3); F8_RE_B(4); F8_RE_B(5); F8_RE_B(6); F8_RE_B(7);
            F8_RE_A(0); F8_RE_A(1); F8_RE_A(2); F8_RE_A(3); F8_RE_A(4); F8_RE_A(5); F8_RE_A(6); F8_RE_A(7);
#undef F8_RE_A
#undef F8_RE_B
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // the next tile accumulates from zero (the K-step body has no "C = 0" variant, see f8_step)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        F8_PIN8(0); F8_PIN8(1); F8_PIN8(2); F8_PIN8(3); F8_PIN8(4); F8_PIN8(5); F8_PIN8(6); F8_PIN8(7);
#undef F8_PIN8
        vid += G;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // in-flight DMA writes / reads must not outlive the workgroup's LDS allocation
}

}  // namespace

static inline int f8_groups(int R) { return (R + 255) / 256 * 4; }      // mx_groups() of fp8.hip

static bool f8_fast_takes(const avllm_gemm_f8_desc* d) {
    const int off = av_knob(AV_KNOB_F8_FAST) == 0;
    const int xtiles = av_cdiv(d->M, TM) * av_cdiv(d->N, TN);
    const bool aligned = (d->Cq || (uintptr_t)d->C % 16 == 0) && (!d->bias || (uintptr_t)d->bias % 16 == 0) && (!d->R || (uintptr_t)d->R % 16 == 0) && d->N % 8 == 0 &&
                         ((uintptr_t)d->A % 16 == 0) && ((uintptr_t)d->B % 16 == 0);
    const bool fits = (double)d->M * (double)d->lda < 4.0e9 && (double)d->N * (double)d->ldb < 4.0e9 && d->lda < (1 << 24) && d->ldb < (1 << 24);
    const bool qok = !d->Cq || (d->SCq && !d->R && d->N % 32 == 0 && d->ldcq % 8 == 0 && (uintptr_t)d->Cq % 8 == 0);
    return !(off || d->M <= 128 || xtiles < 64 || d->K < 2 * KB || !aligned || !fits || !qok);
}
extern "C" int avllm_gemm_f8_takes_quantised_output(const avllm_gemm_f8_desc* d) { return d && d->Cq && f8_fast_takes(d) ? 1 : 0; }

int av_gemm_f8_fast(const avllm_gemm_f8_desc* d, hipStream_t st, bool* taken) {
    *taken = false;
    if (!f8_fast_takes(d)) return AV_OK;
    const int xtiles = av_cdiv(d->M, TM) * av_cdiv(d->N, TN);
    static bool attr[64] = {};
    static int ncu[64] = {};
    int dev = 0;
    AV_HIP(hipGetDevice(&dev));
    dev &= 63;
    if (!attr[dev]) {
        AV_HIP(hipFuncSetAttribute((const void*)gemm_f8_wp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS));
        AV_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        attr[dev] = true;
    }
    F8Args g;
    g.A = (const uint8_t*)d->A; g.B = (const uint8_t*)d->B; g.SA = (const uint32_t*)d->SA; g.SB = (const uint32_t*)d->SB;
    g.lda = d->lda; g.ldb = d->ldb; g.K = d->K; g.RBA = f8_groups(d->M); g.RBB = f8_groups(d->N);
#ifdef AVLLM_EXPERIMENT_KNOBS
    g.dbg = av_knob(AV_KNOB_GEMM_DBG);
#else
    g.dbg = 0;
#endif
    g.C = d->C; g.ldc = d->ldc; g.bias = d->bias; g.R = d->R; g.ldr = d->ldr; g.act = d->act; g.M = d->M; g.N = d->N;
    g.Cq = (uint8_t*)d->Cq; g.SCq = (uint32_t*)d->SCq; g.ldcq = d->ldcq;
    hipLaunchKernelGGL(gemm_f8_wp_kernel, dim3(xtiles < ncu[dev] ? xtiles : ncu[dev]), dim3(256), F8_LDS, st, g);
    AV_LAUNCH_CHECK();
    *taken = true;
    return AV_OK;
}
