"""Per-kernel MFMA-pipe utilisation from a rocprofv3 pass `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` (and, optionally, a second
pass with SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU) joined with the kernel-trace durations.
SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 matrix pipes (32 per v_mfma_f32_32x32x16, 16 per 16x16x32: checked
against the instruction counts of the attention launch); GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs.
usage: python tools/pmc_mfma_util.py <pmc_mfma_dir> <pmc_sq_dir|-> <kernel_trace_dir> [out.md]"""
import csv, glob, collections, re, sys


def name_of(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return re.sub(r'^void ', '', n)[:60]


def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        a = acc[(name_of(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


if __name__ == "__main__":
    A = load(sys.argv[1])
    B = load(sys.argv[2]) if sys.argv[2] != "-" else {}
    tr = list(csv.DictReader(open(glob.glob(sys.argv[3] + "/**/*kernel_trace.csv", recursive=True)[0])))
    D = collections.defaultdict(lambda: [0.0, 0])
    for r in tr:
        k = (name_of(r["Kernel_Name"]), int(r["Grid_Size_X"]))
        D[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); D[k][1] += 1
    nfwd = sum(1 for r in tr if ("im2col_kernel" in r["Kernel_Name"] or "im2col_rows_kernel" in r["Kernel_Name"])) or 1
    rows = []
    for k, c in A.items():
        if k not in D or "SQ_VALU_MFMA_BUSY_CYCLES" not in c: continue
        avg = lambda x: x[0] / max(x[1], 1)
        mf, gui, us = avg(c["SQ_VALU_MFMA_BUSY_CYCLES"]), avg(c["GRBM_GUI_ACTIVE"]), D[k][0] / D[k][1] / 1e3
        if mf == 0: continue
        b = B.get(k, {})
        wave = avg(b["SQ_WAVE_CYCLES"]) if "SQ_WAVE_CYCLES" in b else float("nan")
        rows.append((D[k][0] / nfwd / 1e6, k[0], us, gui / 8 / us / 1e3, mf / (gui / 8 * 1024),
                     avg(b["SQ_WAIT_ANY"]) / wave if "SQ_WAIT_ANY" in b else float("nan"),
                     avg(b["SQ_ACTIVE_INST_VALU"]) / wave if "SQ_ACTIVE_INST_VALU" in b else float("nan")))
    rows.sort(reverse=True)
    out = ["| ms / forward | kernel | us / launch | clock GHz (GUI_ACTIVE / 8 / t) | MFMA pipes busy | waves parked (WAIT_ANY / WAVE_CYCLES) | VALU issue (ACTIVE_INST_VALU / WAVE_CYCLES) |", "|---|---|---|---|---|---|---|"]
    for r in rows[:16]:
        out.append(f"| {r[0]:.2f} | `{r[1]}` | {r[2]:.1f} | {r[3]:.2f} | {100 * r[4]:.1f} % | {r[5]:.2f} | {r[6]:.2f} |")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write("# MFMA-pipe utilisation by kernel (8 x 518^2 bf16, N = 1)\n\n`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` and `--pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU` passes of\n`python bench.py --steps 1 --warmup 1 --no-cpu-baseline`, joined with the kernel trace of the same build.  'MFMA pipes busy' = busy cycles of the 1024\nmatrix pipes / (elapsed shader cycles x 1024): the fraction of the dense peak AT THE CLOCK THE KERNEL RAN AT (the chip holds ~2.1 GHz\nunder the attention kernel, ~2.4 GHz under the GEMMs; the 2.5 PFLOP/s figure bench.py divides by is the 2.4 GHz one).  The attention figure\nincludes the 4 of 36 MFMAs per tile that subtract the running max.\n\n" + text + "\n")
