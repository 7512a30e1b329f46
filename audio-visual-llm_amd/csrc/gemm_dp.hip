// Persistent bf16 NT GEMM with TWO co-resident workgroups per CU:  C[M,N] = act(A[M,K] . B[N,K]^T + A2[M,K2] . B2[N,K2]^T + bias[N]) + R
//
// Why a second kernel beside gemm_bf16_wp_kernel (gemm.hip): that one runs ONE wave per SIMD with all 256 accumulators of a 128x128 quadrant, so
// everything that is not the K loop -- the epilogue (a quarter to 42 % of a K = 768 tile: profiles/r03_gemm_tile_stamps.txt), the refill after
// it, every barrier wait -- leaves the matrix pipe idle.  Here a workgroup owns a 256 x 128 tile (4 waves, 128 x 64 = 128 accumulators each) and
// needs 72 KiB of LDS and <= 256 registers per lane, so two workgroups share a CU: whenever one of them is in its epilogue, waits at a barrier or
// for operands, the other one's MFMAs fill the pipe.  The two pipelines are independent (no cross-workgroup synchronisation): they drift
// out of phase by themselves and are started half a tile apart.
//
//   * K-steps of 32 (one v_mfma_f32_16x16x32_bf16 slab), a ring of three 24-KiB stages per workgroup: A image [256 rows][64 B], B image
//     [128 rows][64 B], filled by global_load_lds_dwordx4 (1 KiB = 16 rows per wave instruction, 6 instructions per lane and K-step).
//     Sub-step t multiplies the fragments read during sub-step t-1 while it reads those of t+1 and requests stage t+3 into the stage it
//     multiplied... one barrier per sub-step:  [vmcnt: my loads of t+1 landed] [barrier: everybody's did, and everybody finished READING
//     stage t] -> stage t is free for t+3.  Loads run three sub-steps ahead of the MFMAs that use them.
//   * 64-byte rows: chunk position = chunk ^ g(key), g(k) = (-k) & 3, key = (row >> 2) & 3 (activations) / (row >> 3) & 3 (weights, whose fragment
//     rows are remapped as below): every ds_read_b128 of a fragment is bank-conflict free (checked by brute force over the four 16-lane groups
//     of MI355X_MICROARCH.md's LDS table), and the swizzle is applied on the SOURCE address of the DMA.
//   * weights are the MFMA A operand and their fragment rows are interleaved (fragment j of a wave = rows 32 (j >> 1) + 4 (j & 1) + 8 (fr >> 2)
//     + (fr & 3)), so a lane's accumulators of fragments 2p and 2p+1 are the 8 consecutive output columns 32p + 8fq .. +7 of row fr: the epilogue
//     stores 16-byte chunks straight from the accumulators, as the one-wave-per-SIMD kernel does.
//   * two K-steps are written out instruction by instruction as one "macro step" (64 MFMAs; both fragment buffers named at compile time), so
//     K and K2 must be multiples of 64 like everywhere else in avllm_gemm.
//   * lean epilogue only (bf16 out, alpha = 1, no dropout / row remap): the dispatcher in gemm.hip keeps every other call on the older kernels.
#include "common.h"
#include "avllm_internal.h"
#include "gemm_shared.h"
#include <utility>
#include <type_traits>

namespace {

using avg::epilogue_fast8; using avg::xcd_remap; using avg::tile_coords;

constexpr int DTM = 256, DTN = 128, DBK = 32;
constexpr int DOFFB = DTM * DBK * 2;                     // 16 KiB: the weight image of a stage follows the activation image
constexpr int DSTAGE = (DTM + DTN) * DBK * 2;            // 24 KiB
constexpr int DLDS = 3 * DSTAGE;                         // 72 KiB per workgroup, two workgroups per CU

struct DpArgs {
    const bf16 *A, *B, *A2, *B2;
    long lda, ldb, lda2, ldb2;
    int K, K2, M, N;
    bf16* C; long ldc;
    const bf16* bias; const bf16* R; long ldr;
    int act, dbg;
};

#define DP_BOFF(j) ((((j) >> 1) * 32 + ((j) & 1) * 4) * 64)
#define DP_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// One DMA instruction: LDS rows rowbase + (lane >> 2) of an image, 16 bytes per lane.  The source offset -- min(row, last row) * row bytes + swizzled
// chunk -- is built right here from scalars (4 VALU slots between MFMAs cost nothing): twelve precomputed offsets per macro step were the
// registers that pushed the first cut of this kernel into scratch.
#define DP_LD(l4, rowbase, rmax, ld2, vc, base, m0v) do { unsigned t_; \
    asm volatile("v_add_u32 %0, %2, %1\n\tv_min_u32 %0, %3, %0\n\tv_mul_u32_u24 %0, %4, %0\n\tv_add_u32 %0, %5, %0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t" \
                 "global_load_lds_dwordx4 %0, %7" : "=&v"(t_) : "v"(l4), "s"(rowbase), "s"(rmax), "s"(ld2), "v"(vc), "s"(m0v), "s"(base) : "memory"); } while (0)
// the scalars of one requested K-step: last valid row of each operand relative to the tile, row strides in bytes, tile pointers at the step's k offset
struct DpSub { unsigned ma, mb, la2, lb2; const bf16 *pa, *pb; };
// per-lane constants of the DMA: lane >> 2 and the swizzled chunk offsets (weights: one per instruction)
struct DpLane { unsigned l4, cA, cB0, cB1; };

// One instruction slot of a macro step (two K-steps of 32): N = 32 h + n, MFMA n of half h = accumulator (n >> 2, n & 3), then whatever else
// the slot carries.  MODE 0: inner macro step; 1: a tile's first (its first half starts the accumulators from 0); 2: the same right after a
// full-tile epilogue, whose 16 stores may stay in flight behind the counted wait of both halves (vmcnt counts loads and stores in issue order).
template <int N, int MODE>
__device__ __forceinline__ void dp_slot(f32x4 (&acc)[8][4], bf16x8 (&FA)[2][8], bf16x8 (&FB)[2][4], const int (&ra)[2], const int (&rb)[2],
                                        const DpSub (&cx)[2], const DpLane& dl, int rowA, int rowB, const int (&mA)[2], const int (&mB)[2]) {
    constexpr int h = N >> 5, n = N & 31, I = n >> 2, J = n & 3;
    if constexpr (MODE != 0 && h == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[I][J]) : "v"(FB[h][J]), "v"(FA[h][I]));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(FB[h][J]), "v"(FA[h][I]));
    if constexpr (n == 2) { if constexpr (MODE == 2) asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    if constexpr (n == 3) __builtin_amdgcn_s_barrier();
    if constexpr (n >= 4 && n < 8) DP_RD(FB[h ^ 1][n - 4], rb[h], DP_BOFF(n - 4));
    if constexpr (n >= 8 && n < 16) DP_RD(FA[h ^ 1][n - 8], ra[h], (n - 8) * 1024);
    if constexpr (n >= 16 && n < 20) DP_LD(dl.l4, rowA + (n - 16) * 16, cx[h].ma, cx[h].la2, dl.cA, cx[h].pa, mA[h] + (n - 16) * 1024);
    if constexpr (n == 20) DP_LD(dl.l4, rowB, cx[h].mb, cx[h].lb2, dl.cB0, cx[h].pb, mB[h]);
    if constexpr (n == 21) DP_LD(dl.l4, rowB + 16, cx[h].mb, cx[h].lb2, dl.cB1, cx[h].pb, mB[h] + 1024);
    if constexpr (n == 31) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int MODE, int... Ns>
__device__ __forceinline__ void dp_macro(std::integer_sequence<int, Ns...>, f32x4 (&acc)[8][4], bf16x8 (&FA)[2][8], bf16x8 (&FB)[2][4], const int (&ra)[2],
                                         const int (&rb)[2], const DpSub (&cx)[2], const DpLane& dl, int rowA, int rowB, const int (&mA)[2], const int (&mB)[2]) {
    (dp_slot<Ns, MODE>(acc, FA, FB, ra, rb, cx, dl, rowA, rowB, mA, mB), ...);
}

template <bool HAS2>
__global__ __launch_bounds__(256, 2) void gemm_bf16_dp_kernel(DpArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_m = (g.M + DTM - 1) / DTM, tiles_n = (g.N + DTN - 1) / DTN;
    const int ntiles = tiles_m * tiles_n, G = gridDim.x;             // G <= ntiles (dispatcher)
    const int ns1 = g.K / DBK, ns = ns1 + g.K2 / DBK;                // sub-steps per tile: even, >= 4 (dispatcher)
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
    // Per-lane constants of the K loop, derived from the hardware lane id and REBUILT after every epilogue (volatile asm: no hoisting, no common
    // subexpression with the previous copy): kept live across the epilogue they were spilled, and their reload in the first K-step of the
    // next tile put a scratch load -- i.e. a compiler vmcnt(0) -- into the counted-wait pipeline.
    int fa, fb, lane;                                                // this lane's fragment rows inside a stage (chunk fq of the row, swizzled)
    DpLane dl;                                                       // DMA: instruction q of wave w writes LDS rows 64 w + 16 q + (lane >> 2) (activations) / 32 w + 16 q + (lane >> 2) (weights), position lane & 3
    auto lane_consts = [&]() __attribute__((always_inline)) {
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        const int fr = lane & 15, fq = lane >> 4;
        const int gk = (-(fr >> 2)) & 3;
        fa = lds0 + (wr * 128 + fr) * 64 + ((fq ^ gk) << 4);
        fb = lds0 + DOFFB + (wc * 64 + 8 * (fr >> 2) + (fr & 3)) * 64 + ((fq ^ gk) << 4);
        dl.l4 = lane >> 2;
        dl.cA = ((lane & 3) ^ ((-(lane >> 4)) & 3)) << 4;
        dl.cB0 = ((lane & 3) ^ ((-(lane >> 5)) & 3)) << 4; dl.cB1 = dl.cB0 ^ 32;      // g(2 + k) = g(k) ^ 2
    };
    lane_consts();
    const int rowA = wave * 64, rowB = wave * 32;

    // Load context (scalars only): the tile whose K-steps are being requested, three sub-steps ahead of the one being multiplied
    const bf16 *tA1, *tB1;
    int lvid = blockIdx.x, lt = 0, lm0 = 0, ln0 = 0;
    auto tile_of = [&](int vid, int& tm, int& tn) __attribute__((always_inline)) { tile_coords(xcd_remap(vid, ntiles), tiles_m, tiles_n, tm, tn, 0, 0); };
    auto set_ctx = [&](int vid) __attribute__((always_inline)) {
        int tm, tn;
        tile_of(vid, tm, tn);
        lm0 = tm * DTM; ln0 = tn * DTN;
        tA1 = g.A + (long)lm0 * g.lda; tB1 = g.B + (long)ln0 * g.ldb;
        if (g.dbg & 0x800) { tA1 = g.A; tB1 = g.B; }                   // experiment: every tile multiplies the first rows of both operands (every line a TCP miss and, for short K, an L2 hit)
    };
    auto advance = [&]() __attribute__((always_inline)) {            // after sub-step lt of the context tile has been requested
        if (++lt < ns) return;
        if (lvid + G < ntiles) { lvid += G; set_ctx(lvid); lt = 0; }
        else lt = ns - 1;                                            // nothing left: keep re-requesting the last sub-step (never read); the vmcnt arithmetic stays uniform
    };
    auto sub_ctx = [&]() __attribute__((always_inline)) {            // the context's current sub-step
        DpSub c;
        c.ma = g.M - 1 - lm0; c.mb = g.N - 1 - ln0;
        if (g.dbg & 0x40) { c.ma = 0; c.mb = 0; }
        if (g.dbg & 0x400) { c.ma = c.ma < 63 ? c.ma : 63; c.mb = c.mb < 63 ? c.mb : 63; }      // ... only the tile's first 64 rows: TCP misses that hit in L2
        if (g.dbg & 0x100) c.ma = 0;
        if (g.dbg & 0x200) c.mb = 0;                    // experiment: every lane fetches row 0 of its operand (one cache line per instruction, always a hit): wrong numbers, same instruction stream
        c.la2 = (unsigned)g.lda * 2; c.lb2 = (unsigned)g.ldb * 2;
        c.pa = tA1 + (long)lt * DBK; c.pb = tB1 + (long)lt * DBK;
        if constexpr (HAS2) {
            if (lt >= ns1) {
                c.la2 = (unsigned)g.lda2 * 2; c.lb2 = (unsigned)g.ldb2 * 2;
                c.pa = g.A2 + (long)lm0 * g.lda2 + (long)(lt - ns1) * DBK; c.pb = g.B2 + (long)ln0 * g.ldb2 + (long)(lt - ns1) * DBK;
            }
        }
        return c;
    };
    set_ctx(lvid);
    {   // the two workgroups of a CU start half a tile apart (dbg bits 4..5: 1 = no stagger, 2 = by workgroup parity instead of by grid half)
        const int how = (g.dbg >> 4) & 3;
        const bool late = how == 2 ? (blockIdx.x & 1) : (2 * (int)blockIdx.x >= G);
        if (how != 1 && late && ntiles >= 2 * G) {
            const int n = (ns * 1100 + 8000) / (2 * 6400);
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(100);
        }
    }
    const int mA0 = lds0 + wave * 4096, mB0 = lds0 + DOFFB + wave * 2048;
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
        const DpSub c = sub_ctx();
#pragma unroll
        for (int q = 0; q < 4; ++q) DP_LD(dl.l4, rowA + q * 16, c.ma, c.la2, dl.cA, c.pa, mA0 + s * DSTAGE + q * 1024);
        DP_LD(dl.l4, rowB, c.mb, c.lb2, dl.cB0, c.pb, mB0 + s * DSTAGE);
        DP_LD(dl.l4, rowB + 16, c.mb, c.lb2, dl.cB1, c.pb, mB0 + s * DSTAGE + 1024);
        advance();
    }
    f32x4 acc[8][4];
    bf16x8 FA[2][8], FB[2][4];
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int s0 = 0, s1 = DSTAGE, s2 = 2 * DSTAGE;                        // stage of the sub-step being multiplied, of the next one, of the one after
#define DP_REREAD() do { const int a0_ = fa + s0, b0_ = fb + s0; \
        DP_RD(FB[0][0], b0_, DP_BOFF(0)); DP_RD(FB[0][1], b0_, DP_BOFF(1)); DP_RD(FB[0][2], b0_, DP_BOFF(2)); DP_RD(FB[0][3], b0_, DP_BOFF(3)); \
        DP_RD(FA[0][0], a0_, 0); DP_RD(FA[0][1], a0_, 1024); DP_RD(FA[0][2], a0_, 2048); DP_RD(FA[0][3], a0_, 3072); \
        DP_RD(FA[0][4], a0_, 4096); DP_RD(FA[0][5], a0_, 5120); DP_RD(FA[0][6], a0_, 6144); DP_RD(FA[0][7], a0_, 7168); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
    using Seq = std::make_integer_sequence<int, 64>;
    auto mstep = [&](auto modec) __attribute__((always_inline)) {
        constexpr int MODE = decltype(modec)::value;
        DpSub cx[2];
        cx[0] = sub_ctx(); advance();
        cx[1] = sub_ctx(); advance();
        // half 0: reads stage s1, requests into s0; half 1: reads s2, requests into s1
        const int ra[2] = {fa + s1, fa + s2}, rb[2] = {fb + s1, fb + s2};
        const int mA[2] = {mA0 + s0, mA0 + s1}, mB[2] = {mB0 + s0, mB0 + s1};
        dp_macro<MODE>(Seq{}, acc, FA, FB, ra, rb, cx, dl, rowA, rowB, mA, mB);
        const int t = s0; s0 = s2; s2 = s1; s1 = t;
    };
    bool stores_in_flight = false;                                   // the previous tile's epilogue issued all of its 16 stores and nothing else
    const int nm = ns >> 1;
    for (int vid = blockIdx.x; vid < ntiles; vid += G) {
        // The tile's first fragments are read here (stage s0 was complete before the last barrier), not carried through the previous epilogue --
        // and inside each branch: read before the branch they reached the two instantiations in different registers by way of scratch
        if (stores_in_flight) { DP_REREAD(); mstep(std::integral_constant<int, 2>{}); }
        else { DP_REREAD(); mstep(std::integral_constant<int, 1>{}); }
#pragma unroll 1
        for (int t = 1; t < nm; ++t) mstep(std::integral_constant<int, 0>{});
        // result latency of the last MFMAs (invisible to the compiler's hazard recogniser): nops, and every accumulator named as an in/out
        // operand so that compiler-generated readers stay behind them
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#define DP_PIN4(I) asm volatile("" : "+a"(acc[I][0]), "+a"(acc[I][1]), "+a"(acc[I][2]), "+a"(acc[I][3]))
        DP_PIN4(0); DP_PIN4(1); DP_PIN4(2); DP_PIN4(3); DP_PIN4(4); DP_PIN4(5); DP_PIN4(6); DP_PIN4(7);
#undef DP_PIN4
        // lane (fr, fq), row block i, column pair p: the 8 consecutive output columns 32p + 8fq .. +7 of row 16i + fr
        int tm, tn;
        tile_of(vid, tm, tn);
        int el;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
        const int m0 = tm * DTM + wr * 128 + (el & 15), n = tn * DTN + wc * 64 + (el >> 4) * 8;
        const bool full = (tm + 1) * DTM <= g.M && (tn + 1) * DTN <= g.N && !(g.dbg & 3);      // wave-uniform: every lane stores all 16 chunks
        stores_in_flight = full;
        float b[2][8];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int c = 0; c < 8; ++c) b[p][c] = 0.f;
            if (g.bias && n + 32 * p < g.N) load_f<8>(g.bias + n + 32 * p, b[p]);
        }
        bf16* const cp0 = g.C + (long)m0 * g.ldc + n;
        const bf16* const rp0 = g.R ? g.R + (long)m0 * g.ldr + n : nullptr;
        const long ldc = g.ldc, ldr = g.R ? g.ldr : 0;
        // edge tiles (and activation + residual, which no model call makes): per-chunk bounds tests
        auto run = [&](auto actc) __attribute__((always_inline)) {
            constexpr int ACT = decltype(actc)::value;
#pragma clang loop unroll(full)
            for (int i = 0; i < 8; ++i) {
                const bool mrow = m0 + i * 16 < g.M;
#pragma clang loop unroll(full)
                for (int p = 0; p < 2; ++p) {
                    asm volatile("" : "+a"(acc[i][2 * p]), "+a"(acc[i][2 * p + 1]));
                    if (mrow && n + 32 * p < g.N && !(g.dbg & 1))
                        epilogue_fast8<ACT>(acc[i][2 * p], acc[i][2 * p + 1], b[p], g.bias != nullptr, rp0 ? rp0 + (i * 16) * ldr + 32 * p : nullptr,
                                            cp0 + (i * 16) * ldc + 32 * p);
                }
            }
        };
        // full tile, no residual: 8 accumulator reads, the bias adds, the activation, 4 packed converts and one 16-byte store per chunk
        auto run_full = [&](auto actc, auto hasbc) __attribute__((always_inline)) {
            constexpr int ACT = decltype(actc)::value;
            constexpr bool HASB = decltype(hasbc)::value;
#pragma clang loop unroll(full)
            for (int i = 0; i < 8; ++i) {
                bf16* const cpi = cp0 + (long)(i * 16) * ldc;
#pragma clang loop unroll(full)
                for (int p = 0; p < 2; ++p) {
                    asm volatile("" : "+a"(acc[i][2 * p]), "+a"(acc[i][2 * p + 1]));      // no hoisting of all accumulator reads above the variant branch
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
        // full tile + residual: all 16 residual chunks of the lane requested in one burst into the (dead) fragment registers, consumed in issue
        // order behind one constant vmcnt(15): 15 - c younger loads + c stores behind chunk c.  asm loads with explicit waits (compiler-visible
        // loads consumed later reach the K loop as vmcnt(0) per K-step); every load is issued before the first store (in-place residual).
        auto run_res_full = [&](auto hasbc) __attribute__((always_inline)) {
            constexpr bool HASB = decltype(hasbc)::value;
            u32x4 rr[8][2];
#pragma clang loop unroll(full)
            for (int i = 0; i < 8; ++i) {
                const bf16* rp = rp0 + (long)(i * 16) * ldr;
                asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64" : "=&v"(rr[i][0]), "=&v"(rr[i][1]) : "v"(rp) : "memory");
            }
#pragma clang loop unroll(full)
            for (int i = 0; i < 8; ++i) {
                bf16* const cpi = cp0 + (long)(i * 16) * ldc;
#pragma clang loop unroll(full)
                for (int p = 0; p < 2; ++p) {
                    u32x4& x = rr[i][p];
                    asm volatile("s_waitcnt vmcnt(15)" : "+v"(x) :: "memory");
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
        if (full && !g.R && g.act == AV_ACT_NONE && !hb) run_full(std::integral_constant<int, AV_ACT_NONE>{}, no);
        else if (full && !g.R && g.act == AV_ACT_NONE && hb) run_full(std::integral_constant<int, AV_ACT_NONE>{}, yes);
        else if (full && !g.R && g.act == AV_ACT_GELU && hb) run_full(std::integral_constant<int, AV_ACT_GELU>{}, yes);
        else if (full && !g.R && g.act == AV_ACT_QUICK_GELU && hb) run_full(std::integral_constant<int, AV_ACT_QUICK_GELU>{}, yes);
        else if (full && g.R && g.act == AV_ACT_NONE && hb) run_res_full(yes);
        else if (full && g.R && g.act == AV_ACT_NONE) run_res_full(no);
        else if (g.act == AV_ACT_NONE) run(std::integral_constant<int, AV_ACT_NONE>{});
        else if (g.act == AV_ACT_GELU) run(std::integral_constant<int, AV_ACT_GELU>{});
        else if (g.act == AV_ACT_QUICK_GELU) run(std::integral_constant<int, AV_ACT_QUICK_GELU>{});
        else run(std::integral_constant<int, AV_ACT_SILU>{});
        lane_consts();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // in-flight DMA writes must not outlive the workgroup's LDS allocation
}

struct DpDevState { int ncu = 0; bool attr = false; };
DpDevState g_dp_dev[64];

}  // namespace

// The calls this kernel takes: what gemm.hip calls the lean epilogue (bf16, alpha = 1, no dropout / row remap / broadcast residual, 16-byte aligned
// rows), K and K2 multiples of 64 with at least two macro steps, operands within 32-bit tile-relative byte offsets.
bool av_gemm_dp_ok(const avllm_gemm_desc* d) {
    if (d->dtype != AV_BF16 || d->out_f32 || d->alpha != 1.f || d->drop_p > 0.f || d->a_drop_p > 0.f || d->g_in > 0 || d->r_mod > 0) return false;
    if (d->K % 64 || d->K2 % 64 || d->K + d->K2 < 128 || d->N % 8 || d->M <= 128) return false;
    if ((uintptr_t)d->C % 16 || (d->ldc * 2) % 16 || (d->bias && (uintptr_t)d->bias % 16) || (d->R && ((uintptr_t)d->R % 16 || d->ldr % 8))) return false;
    const long ldA = d->lda > d->lda2 ? d->lda : d->lda2, ldB = d->ldb > d->ldb2 ? d->ldb : d->ldb2;
    return ldA * 2 < (1l << 24) && ldB * 2 < (1l << 24);             // __umul24 row offsets
}

int av_gemm_dp(const avllm_gemm_desc* d, hipStream_t st, int dbg) {
    AV_CHECK_ARG(av_gemm_dp_ok(d), "gemm (two-workgroup kernel): call outside its lean form");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    DpDevState* s = &g_dp_dev[dev & 63];
    if (!s->ncu && hipDeviceGetAttribute(&s->ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) s->ncu = 256;
    if (!s->attr) {
        AV_HIP(hipFuncSetAttribute((const void*)gemm_bf16_dp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DLDS));
        AV_HIP(hipFuncSetAttribute((const void*)gemm_bf16_dp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DLDS));
        s->attr = true;
    }
    DpArgs g;
    g.A = (const bf16*)d->A; g.B = (const bf16*)d->B; g.A2 = (const bf16*)d->A2; g.B2 = (const bf16*)d->B2;
    g.lda = d->lda; g.ldb = d->ldb; g.lda2 = d->lda2; g.ldb2 = d->ldb2; g.K = d->K; g.K2 = d->K2; g.M = d->M; g.N = d->N;
    g.C = (bf16*)d->C; g.ldc = d->ldc; g.bias = (const bf16*)d->bias; g.R = (const bf16*)d->R; g.ldr = d->ldr; g.act = d->act; g.dbg = dbg;
    const int per_cu = (dbg & 0x80) ? 1 : 2;                         // experiment: one workgroup per CU
    const int ntiles = av_cdiv(d->M, DTM) * av_cdiv(d->N, DTN), grid = ntiles < per_cu * s->ncu ? ntiles : per_cu * s->ncu;
    if (d->K2 > 0) hipLaunchKernelGGL(gemm_bf16_dp_kernel<true>, dim3(grid), dim3(256), DLDS, st, g);
    else hipLaunchKernelGGL(gemm_bf16_dp_kernel<false>, dim3(grid), dim3(256), DLDS, st, g);
    return AV_OK;
}
