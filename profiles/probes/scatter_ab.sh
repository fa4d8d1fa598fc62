#!/bin/bash
# Diagnostic builds of the scatter experiment (never the shipped library): ab/libmmgnn_<n>.so with -DMMG_SR_ABL=<n>
# (bits: see the comment in front of k_scatter_roles in csrc/aggregate.hip; a trailing "s" adds -DMMG_STAMPS for
# scatter_roles_stamps.py).  Usage: profiles/probes/scatter_ab.sh 0 3 16 16384 0s ...   then
#   MMG_AB_LIB=ab/libmmgnn_<n>.so python profiles/probes/scatter_time.py 100 128 400
set -e
cd "$(dirname "$0")/../../multi-modal-gnn_amd/csrc"
mkdir -p ../../ab
for v in "$@"; do
  n=${v%s}; extra=""
  if [ "$n" != "$v" ]; then extra="-DMMG_STAMPS"; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -DMMG_SR_ABL=$n $extra -c aggregate.hip -o ../../ab/aggregate_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/libmmgnn_$v.so api.o csr.o ../../ab/aggregate_$v.o gemm.o elementwise.o pairs.o evalred.o optim.o small.o
done
