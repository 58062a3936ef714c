#!/bin/bash
# The r02 evidence set, one command on the GPU box: bench lines (C2 with cpu_baseline + north-star leg; C3), rocprofv3 kernel trace +
# stats of the C2 workload, and the PMC passes (HBM traffic; matrix-pipe busy; wave-cycle split), each its own run as the guides
# prescribe.  usage: bash tools/collect_r02_evidence.sh [stage ...]   (stages: bench prof pmc c3; default all)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
stages=${@:-bench prof pmc c3}
B="python3 $R/bench.py"
for st in $stages; do
case $st in
bench)
  timeout -k 10 500 $B --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?";;
prof)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/prof.log 2>&1; echo "prof rc $?";;
pmc)
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-north-star > $O/pmc_fetch.log 2>&1; echo "fetch rc $?"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-north-star > $O/pmc_write.log 2>&1; echo "write rc $?"
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-north-star > $O/pmc_mfma.log 2>&1; echo "mfma rc $?"
  timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-north-star > $O/pmc_sq.log 2>&1; echo "sq rc $?";;
c3)
  timeout -k 10 300 $B --steps 3 --warmup 1 --views-per-gpu 32 --priors --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc $?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- $B --steps 1 --warmup 1 --views-per-gpu 32 --priors --no-cpu-baseline > $O/prof_c3.log 2>&1; echo "prof c3 rc $?";;
esac
done
cd $R
ls $O
