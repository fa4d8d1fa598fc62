#!/bin/bash
# Round 4: every profile and bench record kept under profiles/ (run on the GPU box through gpurun; the small files land in
# gpurun_out/r4_<tag>/keep/ and gpurun_out/r4_final/ and are copied to profiles/ afterwards).
#   usage: profiles/collect_all_r4.sh [tags ...]      default: eicu_x100_d128 eicu_x100_d256 mimic_x100_d128 eicu_x1_d128
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r4_final
TAGS=${*:-"eicu_x100_d128 eicu_x100_d256 mimic_x100_d128 eicu_x1_d128"}
args_of() {
  case $1 in
    eicu_x100_d128) echo "";;
    eicu_x100_d256) echo "--dim 256";;
    mimic_x100_d128) echo "--shape mimic";;
    eicu_x1_d128) echo "--scale 1 --cpu-scale 1";;
  esac
}
for t in $TAGS; do
  a=$(args_of $t)
  echo "##### $t ($a)"
  bash profiles/collect_r4.sh $t $a 2>&1 | grep -v "^rc=0" | tail -3
  extra="--no-strong-x1000"
  [ "$t" = "eicu_x100_d128" ] && extra=""
  [ "$t" = "eicu_x100_d128" ] || extra="$extra --no-cpu-baseline"
  timeout -k 10 900 python3 bench.py $a $extra > gpurun_out/r4_final/bench_$t.json 2> gpurun_out/r4_final/bench_$t.err
  echo "bench rc $?  $(python3 -c "import json;d=json.load(open('gpurun_out/r4_final/bench_$t.json'));print(d['ms_per_step'], d['roofline']['frac'])" 2>&1)"
done
