#!/bin/bash
# The r03 evidence set, one command on ONE GPU box (boxes of the pool differ by 5-10 %: every A/B below is same-box): bench lines (C2
# with cpu_baseline + parity + north-star leg; C3; the C5 flag set), rocprofv3 kernel trace + stats of the C2 workload, the PMC passes
# (HBM traffic; matrix-pipe busy; wave-cycle split), each its own run as the guides prescribe, the in-kernel clock stamps of the
# cross-view attention kernels (diagnostic build), the MFMA-shape / bare-MFMA micro loops and the attention A/B.
# usage: bash tools/collect_r03_evidence.sh [stage ...]   (stages: bench prof pmc c3 c5 attn; default all)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
stages=${@:-bench prof pmc c3 c5 attn}
B="python3 $R/bench.py"
for st in $stages; do
case $st in
bench)
  timeout -k 10 700 $B --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?";;
prof)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-parity > $O/prof.log 2>&1; echo "prof rc $?";;
pmc)
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-north-star --no-parity > $O/pmc_fetch.log 2>&1; echo "fetch rc $?"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-north-star --no-parity > $O/pmc_write.log 2>&1; echo "write rc $?"
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-north-star --no-parity > $O/pmc_mfma.log 2>&1; echo "mfma rc $?"
  timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-north-star --no-parity > $O/pmc_sq.log 2>&1; echo "sq rc $?";;
c3)
  timeout -k 10 300 $B --steps 3 --warmup 1 --views-per-gpu 32 --priors --no-cpu-baseline --no-parity > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc $?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- $B --steps 1 --warmup 1 --views-per-gpu 32 --priors --no-cpu-baseline --no-parity > $O/prof_c3.log 2>&1; echo "prof c3 rc $?";;
c5)
  timeout -k 10 300 $B --steps 3 --warmup 1 --dtype f16 --gs --views-per-gpu 32 --no-cpu-baseline --no-north-star > $O/bench_c5flags_32v.json 2> $O/bench_c5flags_32v.err; echo "c5 rc $?";;
attn)
  cd $R
  WM_HIP_LIB=$R/hunyuanworld-mirror_amd/libwm_hip_stamps.so timeout -k 10 300 python3 tools/attn_stamps.py bf16 8 32 > $O/attn_stamps_bf16.jsonl 2>/dev/null; echo "stamps rc $?"
  WM_HIP_LIB=$R/hunyuanworld-mirror_amd/libwm_hip_stamps.so timeout -k 10 300 python3 tools/attn_stamps.py f16 8 32 > $O/attn_stamps_f16.jsonl 2>/dev/null
  timeout -k 10 300 python3 tools/attn_loop_shapes.py 4 > $O/attn_loop_shapes.jsonl 2>/dev/null; echo "micro rc $?"
  CASES=global_8v,global_7v,global_32v REPS=3 timeout -k 10 300 python3 tools/bench_attn_v4.py bf16 3 7 8 > $O/attn_ab_bf16.jsonl 2>/dev/null; echo "ab rc $?"
  CASES=global_8v,global_32v REPS=3 timeout -k 10 300 python3 tools/bench_attn_v4.py f16 3 8 > $O/attn_ab_f16.jsonl 2>/dev/null
  cd /tmp;;
esac
done
cd $R
ls $O
