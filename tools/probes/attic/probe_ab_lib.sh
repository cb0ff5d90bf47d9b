#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# A/B on ONE box: the library of the last commit (tools/probes/hip/lib_prev.so, built by hand from `git show HEAD:...`) against
# the working tree's, per-family device times of a profiled 256-scene solve and of a C4 solve
R=$GRAFT_REPO_ROOT; cd $R
for l in prev product prev product; do
  if [ $l = product ]; then unset PTZCALIB_LIB; else export PTZCALIB_LIB=$R/tools/probes/hip/lib_$l.so; fi
  echo "== $l"; timeout 300 python tools/probes/probe_timing.py ${SIZES:-256} 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['dev_ms'], d['it_per_s'], d['profile_ms'])"
done
if [ "${C4:-0}" = "1" ]; then for l in prev product; do
  if [ $l = product ]; then unset PTZCALIB_LIB; else export PTZCALIB_LIB=$R/tools/probes/hip/lib_$l.so; fi
  echo "== C4 $l"; PTZ_BA_STREAMS=1 timeout 600 python tools/probes/probe_c4_families.py 1000 2>&1 | grep '^{' | cut -c1-600
done; fi
