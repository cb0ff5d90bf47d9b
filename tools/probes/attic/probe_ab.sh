#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# A/B of library builds on ONE box (boxes of the pool differ by up to 10 % on memory-bound kernels): tools/probes/hip/lib_<tag>.so
# ORDERS: elimination orders to run the current build with (PTZ_BA_ORDER), default "nd natural"
R=$GRAFT_REPO_ROOT; T=${1:-ab}; shift || true
mkdir -p $R/gpurun_out/$T; cd $R
for rep in 1 2; do
for lib in "$@"; do
  export PTZCALIB_LIB=$R/tools/probes/hip/lib_$lib.so
  echo "== $lib" | tee -a gpurun_out/$T/timing.txt
  timeout 300 python tools/probes/probe_timing.py ${SIZES:-1 256} 2>&1 | grep '^{' | tee -a gpurun_out/$T/timing.txt
done
unset PTZCALIB_LIB
for o in ${ORDERS:-nd natural}; do
  echo "== current $o" | tee -a gpurun_out/$T/timing.txt
  PTZ_BA_ORDER=$o timeout 300 python tools/probes/probe_timing.py ${SIZES:-1 256} 2>&1 | grep '^{' | tee -a gpurun_out/$T/timing.txt
done; done
