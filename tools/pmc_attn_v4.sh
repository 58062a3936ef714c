#!/bin/bash
# PMC passes over the cross-view attention micro-benchmark (32 views), attn_v3 (attn_qb 7) and attn_v4 (8).
# usage (GPU box): bash tools/pmc_attn_v4.sh <outdir>
set -e
out=${1:-gpurun_out/pmc_v4}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export CASES=global_32v REPS=1 PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -- python3 $R/tools/bench_attn_v4.py bf16 7 8 > $R/$out/trace.log 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $R/$out/pmc1 -- python3 $R/tools/bench_attn_v4.py bf16 7 8 > $R/$out/pmc1.log 2>&1 || true
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/$out/pmc2 -- python3 $R/tools/bench_attn_v4.py bf16 7 8 > $R/$out/pmc2.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$out/pmc3 -- python3 $R/tools/bench_attn_v4.py bf16 7 8 > $R/$out/pmc3.log 2>&1 || true
cd $R
python3 - <<PY
import csv, glob, collections
for d in ("pmc1", "pmc2", "pmc3"):
    fs = glob.glob("$out/" + d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counters (see", d + ".log)"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "attn" not in k: continue
        k = "v3" if "attn_v3" in k else ("v4" if "attn_v4" in k else ("general" if "attn_fwd" in k else "combine"))
        g = int(r["Grid_Size"])
        a = acc[(k, g)][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, c in sorted(acc.items()):
        print(d, k, {n: round(v[0] / v[1]) for n, v in c.items()})
PY
