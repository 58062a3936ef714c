#!/bin/bash
# round 4, VERDICT item 1: GEMM timeline (stamps build), same-box yardsticks, and HEAD vs the round-2 tree on one box
set -e
mkdir -p gpurun_out/r04
WM_HIP_LIB=$PWD/hunyuanworld-mirror_amd/libwm_hip_stamps.so python tools/gemm_timeline.py stamps > gpurun_out/r04/gemm_stamps.jsonl 2> gpurun_out/r04/gemm_stamps.err
python tools/gemm_timeline.py yard > gpurun_out/r04/gemm_yard.jsonl 2> gpurun_out/r04/gemm_yard.err
if [ -d .ab/r02 ]; then
  for i in 1 2; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star --no-parity > gpurun_out/r04/ab_head_$i.json 2> gpurun_out/r04/ab_head_$i.err
    (cd .ab/r02 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star) > gpurun_out/r04/ab_r02_$i.json 2> gpurun_out/r04/ab_r02_$i.err
  done
fi
echo done
