#!/bin/bash
# The r04 evidence set, one command on ONE GPU box (boxes of the pool differ by 5-10 %: every A/B is same-box): bench lines (C2 with
# cpu_baseline + parity + north-star leg; C3; the C5 flag set), rocprofv3 kernel trace + stats of the C2 workload, the PMC passes (HBM
# traffic; matrix-pipe busy; wave-cycle split), each its own run as the guides prescribe, the GEMM timeline / yardsticks / one-part-removed
# builds / schedule A/B, the forward A/B against round 3's GEMM, and the sink-shaped attention rows.
# usage: bash tools/collect_r04_evidence.sh [stage ...]   (stages: bench prof pmc c3 c5 gemm ab heads spike; default all)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
stages=${@:-bench prof pmc c3 c5 gemm ab heads spike}
B="python3 $R/bench.py"
S=$R/hunyuanworld-mirror_amd/libwm_hip_stamps.so
for st in $stages; do
case $st in
bench)
  timeout -k 10 700 $B --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?";;
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
gemm)
  cd $R
  WM_HIP_LIB=$S timeout -k 10 300 python3 tools/gemm_timeline.py stamps > $O/gemm_stamps.jsonl 2>/dev/null; echo "stamps rc $?"
  WM_HIP_LIB=$S timeout -k 10 300 python3 tools/gemm_timeline.py exp 11008 > $O/gemm_exp.jsonl 2>/dev/null; echo "exp rc $?"
  timeout -k 10 300 python3 tools/gemm_timeline.py yard > $O/gemm_yard.jsonl 2>/dev/null; echo "yard rc $?"
  timeout -k 10 300 python3 tools/gemm_timeline.py sched > $O/gemm_sched.jsonl 2>/dev/null; echo "sched rc $?"
  cd /tmp;;
ab)
  cd $R
  timeout -k 10 300 python3 tools/ab_forward.py 8 default: r03gemm:gemm_sched=0 > $O/ab_forward_8v.jsonl 2>/dev/null; echo "ab rc $?"
  cd /tmp;;
heads)
  cd $R
  timeout -k 10 300 python3 tools/ab_forward.py 8 default: direct:tconv=0,up1_gather=0 default2: tconv_only:up1_gather=0 default3: up1_only:tconv=0 direct2:tconv=0,up1_gather=0 > $O/ab_head_rewrites.jsonl 2>/dev/null; echo "heads ab rc $?"
  timeout -k 10 200 python3 tools/bench_upconv_parts.py > $O/upconv_parts.jsonl 2>/dev/null; echo "parts rc $?"
  timeout -k 10 200 python3 tools/bench_conv_gemm.py > $O/dpt_conv_forms.jsonl 2>/dev/null; echo "forms rc $?"
  cd /tmp;;
spike)
  cd $R
  for d in bf16 f16; do SPIKE=1 CASES=global_8v,global_16v,global_32v REPS=3 timeout -k 10 300 python3 tools/bench_attn_v4.py $d 3 8 >> $O/attn_spike.jsonl 2>/dev/null; done; echo "spike rc $?"
  CASES=global_8v,global_32v,frame_8x1376 REPS=3 timeout -k 10 300 python3 tools/bench_attn_v4.py bf16 3 7 8 > $O/attn_ab_bf16.jsonl 2>/dev/null
  cd /tmp;;
esac
done
cd $R
ls $O
