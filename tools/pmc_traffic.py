"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, guides/MI355X_MICROARCH.md §HBM) of
`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-north-star` into profiles/r02_traffic_n1.json: HBM-side bytes per launch
for each kernel row of bench.py's `roofline.kernels` table (and the gemm / dpt_conv classes as wholes).
Usage: python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [out.json]"""
import csv, glob, json, re, sys, collections


def rows_of(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    return [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]


def blocks_of(r):
    wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "256"
    return int(r["Grid_Size"]) // max(int(wg), 1)


def classify(rows):
    """kernel row name per dispatch (None = not a timed row)"""
    fast = lambda n: "attn_v3_kernel" in n or "attn_v4_kernel" in n   # cross-view attention at 8 views (round 3: attn_v4)
    fast_blocks = {blocks_of(r) for r in rows if fast(r["Kernel_Name"])}
    # a fast launch is followed by a launch of the general kernel that re-runs flagged units only (normally none): same block
    # count (256 threads per block behind attn_v3, 512 behind attn_v4)
    rs_plain = sorted({int(r["Grid_Size"]) for r in rows if re.search(r"conv3x3_rs_kernel<\d+, \d+, \d+, \d+, \d+, 0,", r["Kernel_Name"])}, reverse=True)
    out = []
    for r in rows:
        n, g = r["Kernel_Name"], int(r["Grid_Size"])
        k = None
        if fast(n): k = "global_attention"
        elif "attn_fwd_kernel" in n or "attn_sp" in n: k = "global_attention_recheck" if blocks_of(r) in fast_blocks else "frame_dino_attention"
        elif "gemm_pp" in n or "gemm_nt" in n:
            m = re.search(r"gemm_pp2?_kernel<\d+, (\d+)", n)
            e = int(m.group(1)) if m else -1
            k = {6: "gemm_qkv", 3: "gemm_proj_fc2", 2: "gemm_fc1"}.get(e, "gemm_other")
        elif "conv3x3_n32" in n: k = "dpt_output_conv2_tail"
        elif re.search(r"conv3x3_rs_kernel<\d+, \d+, \d+, \d+, \d+, 1,", n): k = "dpt_output_conv1_up"
        elif "conv3x3_rs_kernel" in n and rs_plain and g == rs_plain[0]: k = "dpt_conv3x3_level4x"
        elif "conv3x3_rs_kernel" in n and len(rs_plain) > 1 and g == rs_plain[1]: k = "dpt_conv3x3_level2x"
        elif "conv" in n: k = "dpt_conv_other"
        out.append(k)
    return out


def load(d, counter):
    rows = rows_of(d, counter)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r, k in zip(rows, classify(rows)):
        if k:
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


if __name__ == "__main__":
    fd, wd = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/r02_traffic_n1.json"
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    for cls, pre in (("gemm", "gemm_"), ("dpt_conv", "dpt_")):
        for D in (F, W):
            ks = [k for k in D if k.startswith(pre)]
            D[cls] = [sum(D[k][0] for k in ks), sum(D[k][1] for k in ks)]
    res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 1 --warmup 1 "
                   "--no-cpu-baseline --no-north-star` (8 x 518^2 bf16); values in KiB per dispatch as reported; FETCH_SIZE is doubled (gfx950 "
                   "reports 1/2 of wide coalesced reads, guides/MI355X_MICROARCH.md §HBM); launches_sampled covers the 2 forwards of the run",
           "kernels": {}}
    for k in F:
        n = F[k][1]
        if not n: continue
        fk, wk = F[k][0] / n, W[k][0] / max(W[k][1], 1)
        res["kernels"][k] = {"launches_sampled": n, "fetch_kib_raw": round(fk, 1), "write_kib": round(wk, 1),
                             "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))
