// Pieces shared by the bf16 (gemm.hip) and the block-scaled fp8 (fp8.hip) persistent 256x256 GEMM kernels.
#pragma once
#include "common.h"

namespace avg {

// Lean item of the LDS-staged epilogue for the common case (bf16 output, alpha = 1, no dropout, no row remap, plain residual, tile fully
// inside the matrix): the general epilogue_store8 spends most of its instructions on 64-bit index arithmetic and on uniform branches
// around features these calls do not use.  lo/hi = the two fp32 LDS chunks of this item, b = the thread's bias (zeros without one).
template <int ACT>
__device__ __forceinline__ void epilogue_fast8(const f32x4 lo, const f32x4 hi, const float (&b)[8], bool has_b, const bf16* rp, bf16* cp) {      // rp may alias cp (in-place residual)
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if (has_b) {                                                   // uniform
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += b[i];
    }
    if constexpr (ACT != AV_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = act_apply_fast(v[i], ACT);
    }
    if (rp) {
        const bf16x8 r = *(const bf16x8*)rp;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += (float)r[i];
    }
    store_f<8>(cp, v);
}

// XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous
// range of logical tile ids.  Bijective for any grid size (cdna guide §5, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// gw > 0 (tall shapes, tiles_m > 32): COLUMN GROUPS of gw tile columns, ids run row-major inside a group (tn fastest) and group after group.
// With one contiguous id range per XCD (xcd_remap) an XCD then works inside ONE group for most of the launch: its gw weight panels
// (gw x 256 x K x 2 bytes: 1.2 - 1.6 MB at K = 768) stay L2-resident and each activation panel is fetched once per group and shared by the gw
// CUs that hold its tiles at that moment.  Plain row-major order over all tiles_n columns (gw = 0) cycles the whole weight matrix (4.7 MB for the
// CLIP fc1 > the 4 MiB L2) through every XCD: profiles/r03_pmc_gemm_shapes.txt measured 8.8x (fc1) / 5.3x (qkv) the algorithmic read bytes.
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn, int blocked = 0, int gw = 0) {
    if (gw > 0 && tiles_m > 32 && tiles_n > gw) {
        const int per_group = gw * tiles_m, ngf = tiles_n / gw, full = ngf * per_group;
        if (id < full) {
            const int grp = id / per_group, in = id - grp * per_group;
            tm = in / gw; tn = grp * gw + (in - tm * gw);
        } else {
            const int r = id - full, rem = tiles_n - ngf * gw;
            tm = r / rem; tn = ngf * gw + (r - tm * rem);
        }
        return;
    }
    if (blocked && tiles_m <= 32 && tiles_m % 8 == 0) {
        // 8 x 4 blocks of tiles per 32 consecutive ids (an XCD's share of a round): half the activation panel and four weight panels per XCD
        // and round instead of the whole activation panel and two weight panels -- a third fewer bytes into each L2.  Measured on the
        // Llama shapes (M = 4096): gate|up 604 -> 578 us, d(gate,up) 527 -> 503, fused qkv 341 -> 329, d(lm_head) 734 -> 709; one-round
        // shapes unchanged.  AVLLM_GEMM_DBG bit 2 restores the row-major order.
        const int full = (tiles_n / 4) * 4 * tiles_m;
        if (id < full) {
            const int per_group = 4 * tiles_m, grp = id / per_group, in = id - grp * per_group;
            const int c = in >> 5, w = in & 31;
            tm = c * 8 + (w & 7);
            tn = grp * 4 + (w >> 3);
            return;
        }
        const int r = id - full;
        tm = r % tiles_m; tn = (tiles_n / 4) * 4 + r / tiles_m;
        return;
    }
    if (tiles_m <= 32) { tm = id % tiles_m; tn = id / tiles_m; }      // weights streamed once, all row tiles adjacent
    else               { tn = id % tiles_n; tm = id / tiles_n; }      // activations streamed once
}

}  // namespace avg
