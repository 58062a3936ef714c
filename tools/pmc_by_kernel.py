"""Per-kernel HBM traffic and achieved bandwidth: joins the two rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 correction of
guides/MI355X_MICROARCH.md, WRITE_SIZE) with the kernel-trace durations of the same workload.
usage: python tools/pmc_by_kernel.py <pmc_fetch_dir> <pmc_write_dir> <kernel_trace_dir> [out.md]"""
import csv, glob, collections, re, sys


def name_of(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return re.sub(r'^void ', '', n)[:60]


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = (name_of(r["Kernel_Name"]), int(r["Grid_Size"]))
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


if __name__ == "__main__":
    F, W = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    tr = list(csv.DictReader(open(glob.glob(sys.argv[3] + "/**/*kernel_trace.csv", recursive=True)[0])))
    D = collections.defaultdict(lambda: [0.0, 0])
    for r in tr:
        k = (name_of(r["Kernel_Name"]), int(r["Grid_Size_X"]))
        D[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); D[k][1] += 1
    nfwd = sum(1 for r in tr if ("im2col_kernel" in r["Kernel_Name"] or "im2col_rows_kernel" in r["Kernel_Name"])) or 1
    rows = []
    for k in F:
        if "__amd_rocclr" in k[0]: continue   # the runtime's copy kernels: the weight upload at start-up, not part of a forward
        if k in D and k in W:
            f, w, us = 2 * F[k][0] / F[k][1] * 1024, W[k][0] / W[k][1] * 1024, D[k][0] / D[k][1] / 1e3
            rows.append((D[k][0] / nfwd / 1e6, k, D[k][1] / nfwd, f / 1e6, w / 1e6, us, (f + w) / us / 1e6))
    rows.sort(reverse=True)
    out = ["| ms / forward | kernel | blocks x threads | launches / forward | HBM read MB / launch | HBM write MB / launch | us / launch | achieved HBM TB/s |", "|---|---|---|---|---|---|---|---|"]
    keep = rows[:36] + [r for r in rows[36:] if "im2col" in r[1][0] or "attn" in r[1][0]]
    for r in keep:
        out.append(f"| {r[0]:.2f} | `{r[1][0]}` | {r[1][1]} threads | {r[2]:.0f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} | {r[6]:.2f} |")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write("# HBM traffic and achieved bandwidth by kernel (8 x 518^2 bf16, N = 1)\n\nrocprofv3 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (FETCH_SIZE doubled: gfx950 reports half of wide coalesced reads) joined with the\n`--kernel-trace` durations of the same command (`python3 bench.py --steps 1|3 --warmup 1 --no-cpu-baseline --no-north-star`); HBM peak 8 TB/s.\nThe MFMA-bound kernels (GEMM, attention, conv) sit far below it by design; the HBM-bound ones (LayerNorm, bilinear, combine, copies) are the ones to read against the peak.\n\n" + text + "\n")
