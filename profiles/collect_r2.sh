#!/bin/bash
# Final profile collection of a round (runs on the GPU box): every step under its own timeout, stop at the first kill.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { # name timeout cmd...
  local name=$1 tmo=$2; shift 2
  echo "=== $name ($(date +%T))"
  timeout -k 10 $tmo "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit 1; fi
}
BA="--no-cpu-baseline --no-kernels --no-strong-x1000"
step stats_eager 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c_stats_eager -- python3 $R/bench.py --steps 5 --warmup 2 --no-graph $BA
step stats_graph 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c_stats_graph -- python3 $R/bench.py --steps 40 --warmup 2 $BA
step stats_mimic 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c_stats_mimic -- python3 $R/bench.py --steps 40 --warmup 2 --shape mimic $BA
step stats_x1 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c_stats_x1 -- python3 $R/bench.py --steps 40 --warmup 2 --scale 1 $BA
step pmc_fetch 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BA
step pmc_write 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BA
step pmc_a 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/c_pmc_a -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BA
step pmc_b 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/c_pmc_b -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BA
step pmc_c 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/c_pmc_c -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BA
cd $R
step summarise 120 python3 profiles/summarise_r2.py
exit 0
