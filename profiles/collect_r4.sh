#!/bin/bash
# Profile collection of round 4 (runs on the GPU box): every step under its own timeout, stop at the first kill.
#   usage: profiles/collect_r4.sh <tag> [bench args ...]      e.g.  collect_r4.sh eicu_x100_d128
#                                                                   collect_r4.sh eicu_x100_d256 --dim 256
#                                                                   collect_r4.sh mimic_x100_d128 --shape mimic
# Writes gpurun_out/r4_<tag>/{stats_eager,stats_graph,pmc_*}/ and, through summarise_r4.py, the small files that are kept
# under profiles/ (r4_kernel_stats_<tag>*.txt/csv, r4_step_sequence_<tag>.txt, r4_traffic_<tag>.json, r4_pmc_<tag>.json).
# PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { # name timeout cmd...
  local name=$1 tmo=$2; shift 2
  echo "=== $name ($(date +%T))"
  timeout -k 10 $tmo "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "rc=$rc"
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; tail -5 $O/$name.log; exit 1; fi
}
BA="--no-cpu-baseline --no-kernels --no-strong-x1000 --no-extras $*"
step stats_eager 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_eager -- python3 $R/bench.py --steps 5 --warmup 2 --no-graph $BA
step stats_graph 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_graph -- python3 $R/bench.py --steps 40 --warmup 2 $BA
step pmc_fetch 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BA
step pmc_write 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BA
if [ "${PMC_EXTRA:-1}" = "1" ]; then
step pmc_a 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BA
step pmc_b 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_b -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BA
step pmc_c 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_c -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BA
fi
cd $R
step summarise 120 python3 profiles/summarise_r4.py $TAG
exit 0
