#!/bin/bash
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/p_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
