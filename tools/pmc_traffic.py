"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, guides/MI355X_MICROARCH.md §HBM) of
`python bench.py --steps 1 --warmup 1 --no-cpu-baseline` into profiles/r01_traffic_n1.json (HBM bytes per launch by
kernel class).  Usage: python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [out.json]"""
import csv, glob, json, sys, collections


def klass(name, grid):
    if "gemm" in name: return "gemm"
    if "conv" in name: return "dpt_conv"
    if "attn_fwd" in name or "attn_sp" in name:
        return "global_attention" if grid == GLOBAL_GRID else "frame_dino_attention"
    return None


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    grids = collections.Counter(int(r["Grid_Size"]) for r in rows if "attn" in r["Kernel_Name"])
    global GLOBAL_GRID
    GLOBAL_GRID = min(grids, key=lambda g: grids[g]) if grids else -1  # 24 cross-view launches vs 48 frame/DINO
    for r in rows:
        k = klass(r["Kernel_Name"], int(r["Grid_Size"]))
        if k:
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


if __name__ == "__main__":
    fd, wd = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/r01_traffic_n1.json"
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 1 --warmup 1 "
                   "--no-cpu-baseline` (8 x 518^2 bf16); values in KiB per dispatch as reported; FETCH_SIZE is doubled (gfx950 "
                   "reports 1/2 of wide coalesced reads, guides/MI355X_MICROARCH.md §HBM)", "kernels": {}}
    for k in F:
        n = F[k][1]
        fk, wk = F[k][0] / n, W[k][0] / max(W[k][1], 1)
        res["kernels"][k] = {"launches_sampled": n, "fetch_kib_raw": round(fk, 1), "write_kib": round(wk, 1),
                             "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))
