// Ground truth for v_mfma_scale_f32_16x16x128_f8f6f4 operand layout on gfx950: which (lane, byte) of the A operand meets which
// (lane, byte) of the B operand, and which lane's scale byte applies to which of them.  Prints a compact description.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_scale_probe.hip -o tools/ubench/bin/mfma_scale_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// one wave; a[64][32], b[64][32] bytes; sa[64], sb[64] dwords; out[64][4]
__global__ void k(const uint8_t* a, const uint8_t* b, const int* sa, const int* sb, float* out, int opa, int opb) {
    const int l = threadIdx.x;
    v8i fa, fb;
    for (int r = 0; r < 8; ++r) { fa[r] = ((const int*)a)[l * 8 + r]; fb[r] = ((const int*)b)[l * 8 + r]; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (opa == 0 && opb == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, acc, 0, 0, 0, sa[l], 0, sb[l]);
    else if (opa == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, acc, 0, 0, 1, sa[l], 0, sb[l]);
    else if (opa == 2) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, acc, 0, 0, 2, sa[l], 0, sb[l]);
    else if (opa == 3) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, acc, 0, 0, 3, sa[l], 0, sb[l]);
    else if (opb == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, acc, 0, 0, 0, sa[l], 1, sb[l]);
    else if (opb == 2) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, acc, 0, 0, 0, sa[l], 2, sb[l]);
    else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, acc, 0, 0, 0, sa[l], 3, sb[l]);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}

static uint8_t *da, *db; static int *dsa, *dsb; static float* dout;
static uint8_t ha[2048], hb[2048]; static int hsa[64], hsb[64]; static float ho[256];
static void run(int opa = 0, int opb = 0) {
    hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout, opa, opb);
    hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
}
// D element (row i of the first operand, column j of the second): lane = j + 16 * (i / 4), reg = i % 4
static float D(int i, int j) { return ho[(j + 16 * (i / 4)) * 4 + (i % 4)]; }

int main() {
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 1024);
    const uint8_t ONE = 0x38;      // e4m3 1.0
    for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 0x7f7f7f7f;      // 2^0 in every byte
    // 1. which (group, byte) of A pairs with which (group, byte) of B: A row 0 one-hot, B column 0 one-hot
    printf("K pairing: A(lane group gA, byte bA) meets B(gB, bB)\n");
    int ident = 1;
    for (int gA = 0; gA < 4; ++gA)
        for (int bA = 0; bA < 32; ++bA) {
            memset(ha, 0, sizeof ha); ha[(0 + 16 * gA) * 32 + bA] = ONE;
            int found = -1;
            for (int gB = 0; gB < 4 && found < 0; ++gB) {
                // all 32 bytes of B lane (col 0, group gB) set: find the group first
                memset(hb, 0, sizeof hb); for (int bb = 0; bb < 32; ++bb) hb[(0 + 16 * gB) * 32 + bb] = ONE;
                run();
                if (D(0, 0) == 1.0f) found = gB;
            }
            int fb = -1;
            for (int bB = 0; bB < 32 && found >= 0; ++bB) {
                memset(hb, 0, sizeof hb); hb[(0 + 16 * found) * 32 + bB] = ONE;
                run();
                if (D(0, 0) == 1.0f) { fb = bB; break; }
            }
            if (found != gA || fb != bA) { ident = 0; printf("  A(g%d,b%d) <-> B(g%d,b%d)\n", gA, bA, found, fb); }
        }
    printf("  %s\n", ident ? "identity: same lane group and byte position on both operands" : "NOT identity (pairs listed above)");
    // 2. rows / columns: A lane l holds row l % 16 ?  B lane l holds column l % 16 ?
    memset(ha, 0, sizeof ha); memset(hb, 0, sizeof hb);
    for (int l = 0; l < 64; ++l) { ha[l * 32] = (uint8_t)(0x38 + 8 * 0); }
    for (int r = 0; r < 16; ++r) for (int g = 0; g < 4; ++g) { ha[(r + 16 * g) * 32 + 0] = 0; }
    ha[(5 + 16 * 0) * 32 + 0] = ONE;                        // A: only lane 5 (group 0) byte 0
    for (int c = 0; c < 16; ++c) hb[(c + 16 * 0) * 32 + 0] = ONE;      // B: every column, group 0 byte 0
    run();
    printf("row check: A lane 5 -> nonzero D rows:");
    for (int i = 0; i < 16; ++i) if (D(i, 3) != 0.f) printf(" %d", i);
    printf("\n");
    // 3. scales: set A's scale of ONE lane (row 0, group gs) to 2^1, all K of A row 0 and B col 0 ones -> D(0,0) = 128 + 32 if that lane's
    //    scale applies to exactly its own 32 elements
    memset(ha, 0, sizeof ha); memset(hb, 0, sizeof hb);
    for (int g = 0; g < 4; ++g) for (int bb = 0; bb < 32; ++bb) { ha[(0 + 16 * g) * 32 + bb] = ONE; hb[(0 + 16 * g) * 32 + bb] = ONE; }
    for (int gs = 0; gs < 4; ++gs) {
        for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 0x7f7f7f7f;
        hsa[0 + 16 * gs] = 0x7f7f7f80;                       // byte 0 = 2^1
        run(0, 0);
        printf("scale A lane (row0, group %d) byte0 = 2 with opsel 0: D(0,0) = %g (128 = ignored, 160 = scales its own 32)\n", gs, D(0, 0));
    }
    for (int op = 1; op < 4; ++op) {
        for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 0x7f7f7f7f;
        hsa[0] = 0x7f7f7f7f ^ (0xff << (8 * op)) | (0x80 << (8 * op));      // byte `op` of lane 0 = 2^1
        run(op, 0);
        printf("scale A lane 0 byte %d = 2, opsel_a = %d: D(0,0) = %g\n", op, op, D(0, 0));
        for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 0x7f7f7f7f;
        hsb[0] = 0x7f7f7f7f ^ (0xff << (8 * op)) | (0x80 << (8 * op));
        run(0, op);
        printf("scale B lane 0 byte %d = 2, opsel_b = %d: D(0,0) = %g\n", op, op, D(0, 0));
    }
    // 4. does a scale of a lane in group g also touch other groups?  scale group 1 lane, one-hot K in group 0
    memset(ha, 0, sizeof ha); memset(hb, 0, sizeof hb);
    ha[0] = ONE; hb[0] = ONE;
    for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 0x7f7f7f7f;
    hsa[16] = 0x7f7f7f80;
    run();
    printf("one-hot k in group 0, scale of group-1 lane doubled: D(0,0) = %g (1 = untouched)\n", D(0, 0));
    // 6. WHICH 32 elements does a lane's scale apply to?  A row 0 all ones; B column 0 one-hot at (lane group G, byte position x*16): D(0,0) = 2
    //    iff the A-scale lane doubled covers that element (and the same with the roles swapped)
    for (int which = 0; which < 2; ++which) {
        printf("%s-side: scale lane group g covers 16-byte halves (data lane group G, half x):\n", which ? "B" : "A");
        for (int gs = 0; gs < 4; ++gs) {
            printf("  g=%d:", gs);
            for (int G = 0; G < 4; ++G) for (int x = 0; x < 2; ++x) {
                memset(ha, 0, sizeof ha); memset(hb, 0, sizeof hb);
                uint8_t* full = which ? hb : ha; uint8_t* hot = which ? ha : hb;
                for (int g = 0; g < 4; ++g) for (int bb = 0; bb < 32; ++bb) full[(0 + 16 * g) * 32 + bb] = ONE;
                hot[(0 + 16 * G) * 32 + 16 * x + 3] = ONE;
                for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 0x7f7f7f7f;
                (which ? hsb : hsa)[0 + 16 * gs] = 0x7f7f7f80;
                run();
                if (D(0, 0) == 2.0f) printf(" (G%d,%s)", G, x ? "hi" : "lo");
            }
            printf("\n");
        }
    }
    // 5. every lane's scale: all data ones (D = 128 everywhere); double ONE lane's scale byte and report which D entries move
    for (int which = 0; which < 2; ++which) {
        printf("%s-side scale lanes (opsel 0): lane L doubled -> expected row/col L%%16 gains 32 in every %s\n", which ? "B" : "A", which ? "row" : "column");
        int bad = 0;
        for (int L = 0; L < 64; ++L) {
            for (int i = 0; i < 2048; ++i) ha[i] = hb[i] = ONE;
            for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 0x7f7f7f7f;
            (which ? hsb : hsa)[L] = 0x7f7f7f80;
            run();
            int ok = 1;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                const float exp = 128.f + (((which ? j : i) == (L & 15)) ? 32.f : 0.f);
                if (D(i, j) != exp) ok = 0;
            }
            if (!ok) {
                ++bad;
                printf("  lane %2d (assumed %s %2d, K-block %d): changed entries:", L, which ? "col" : "row", L & 15, L >> 4);
                int shown = 0;
                for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (D(i, j) != 128.f && shown < 6) { printf(" D(%d,%d)=%g", i, j, D(i, j)); ++shown; }
                printf("\n");
            }
        }
        printf("  %d lanes deviate from the assumption\n", bad);
    }
    return 0;
}
