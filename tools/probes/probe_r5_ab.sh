#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# round 5 A/B on ONE box: tools/probes/hip/lib_prev.so (the last commit's library, built by hand from `git archive HEAD`) against the
# working tree's: BA parity subset, then per-family device times of a profiled 256-scene solve and of the C4 solve
R=$GRAFT_REPO_ROOT; cd $R
[ "${TESTS:-1}" = "1" ] && timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 900 -k "${K:-ba_c1 or ba_c2 or medium or global_memory or 6000 or georef or trailing or linearize}" 2>&1 | tail -4
for l in prev product prev product; do
  if [ $l = product ]; then unset PTZCALIB_LIB; else export PTZCALIB_LIB=$R/tools/probes/hip/lib_$l.so; fi
  echo "== $l"; timeout 300 python tools/probes/probe_timing.py ${SIZES:-256} 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['B'], [round(x,2) for x in d['dev_ms']], round(d['it_per_s']), d['profile_ms'])"
done
if [ "${C4:-1}" = "1" ]; then for l in prev product prev product; do
  if [ $l = product ]; then unset PTZCALIB_LIB; else export PTZCALIB_LIB=$R/tools/probes/hip/lib_$l.so; fi
  echo "== C4 $l"; PTZ_BA_STREAMS=1 timeout 600 python tools/probes/probe_c4_families.py 1000 2>&1 | grep '^{' | cut -c1-420
done; fi
