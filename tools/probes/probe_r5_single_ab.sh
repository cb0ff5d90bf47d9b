#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# one rig (C2) and the 64-rig lock step: tools/probes/hip/lib_prev.so against the working tree's library, wall time per LM iteration
R=$GRAFT_REPO_ROOT; cd $R
for l in prev product prev product; do
  if [ $l = product ]; then unset PTZCALIB_LIB; else export PTZCALIB_LIB=$R/tools/probes/hip/lib_$l.so; fi
  echo "== $l"; timeout 300 python tools/probes/probe_run.py 1 8 2>&1 | tail -1
done
if [ "${IBA:-1}" = "1" ]; then for l in prev product; do
  if [ $l = product ]; then unset PTZCALIB_LIB; else export PTZCALIB_LIB=$R/tools/probes/hip/lib_$l.so; fi
  echo "== iba $l"; timeout 600 python tools/probes/probe_iba_batch.py 64 200 2>&1 | grep -E "solo|rigs" | tail -2 | cut -c1-260
done; fi
