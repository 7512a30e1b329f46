#!/usr/bin/env python3
"""Per-SHAPE HBM-side traffic of the GEMM launches: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/gemm_bench.py,
whose shapes run one after the other with a torch fill kernel (out.zero_()) between them -- the dispatch order therefore splits into one
group of GEMM launches per shape, in the order of gemm_bench.SHAPES.  Counter unit = KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md §HBM:
a wide coalesced read stream is tallied at half its bytes), WRITE_SIZE exact.  Algorithmic bytes = one pass over A, B and C.
Usage: pmc_shape_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv>"""
import csv, os, sys
csv.field_size_limit(1 << 30)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def groups(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    out, cur = [], []
    for r in rows:
        if "gemm_" in r["Kernel_Name"]:
            cur.append((r["Kernel_Name"].split("(")[0].split("::")[-1][:40], float(r["Counter_Value"])))
        elif cur:
            out.append(cur); cur = []
    if cur:
        out.append(cur)
    return out


def main():
    MB = int(os.environ.get("ROWS", "4096"))
    shapes = [(MB, 4096, 4096, "llama q/k/v/o"), (MB, 12288, 4096, "llama qkv fused"), (MB, 22016, 4096, "llama gate+up"), (MB, 4096, 11008, "llama down"),
              (MB, 4096, 22016, "llama d(gate,up)"), (MB, 11008, 4096, "llama d(down)"), (MB, 32000, 4096, "lm_head"), (MB, 4096, 32000, "d(lm_head)"),
              (394000, 2304, 768, "clip qkv"), (394000, 768, 768, "clip out"), (394000, 3072, 768, "clip fc1"), (394000, 768, 3072, "clip fc2"),
              (24000, 2304, 768, "whisper qkv"), (24000, 3072, 768, "whisper fc1"), (24000, 768, 3072, "whisper fc2")]
    f, w = groups(sys.argv[1]), groups(sys.argv[2])
    print(f"{'shape':18s} {'M':>7s} {'N':>6s} {'K':>6s}  {'kernel':28s} {'launches':>8s} {'read MiB':>9s} {'alg read':>9s} {'ratio':>6s} {'write MiB':>10s} {'alg write':>10s} {'ratio':>6s}")
    if len(f) != len(shapes) or len(w) != len(shapes):
        print(f"# groups found: fetch {len(f)}, write {len(w)}, shapes {len(shapes)} -- the fill-kernel separator rule did not hold; raw group sizes:", [len(g) for g in f])
    for (M, N, K, tag), gf, gw in zip(shapes, f, w):
        rd = 2 * sum(v for _, v in gf) * 1024 / len(gf) / 2 ** 20
        wr = sum(v for _, v in gw) * 1024 / len(gw) / 2 ** 20
        ar = (M * K + N * K) * 2 / 2 ** 20
        aw = M * N * 2 / 2 ** 20
        print(f"{tag:18s} {M:7d} {N:6d} {K:6d}  {gf[0][0]:28s} {len(gf):8d} {rd:9.1f} {ar:9.1f} {rd / ar:6.2f} {wr:10.1f} {aw:10.1f} {wr / aw:6.2f}")


if __name__ == "__main__":
    main()
