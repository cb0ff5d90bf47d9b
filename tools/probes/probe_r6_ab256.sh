#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# A/B on one box: tools/probes/hip/lib_prev.so against the working tree's library, per-family device times of a profiled 256-scene solve
R=$GRAFT_REPO_ROOT; cd $R
[ "${TESTS:-1}" = "1" ] && timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${K:-c2_parity or batch_matches_single or medium or ragged or c1_parity or many_observations or structure or elimination}" 2>&1 | tail -2
for l in prev product prev product; do
  if [ $l = product ]; then unset PTZCALIB_LIB; else export PTZCALIB_LIB=$R/tools/probes/hip/lib_$l.so; fi
  echo "== $l"; timeout 300 python tools/probes/probe_timing.py ${SIZES:-256} 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['B'], [round(x,2) for x in d['dev_ms']], round(d['it_per_s']), d['profile_ms'])"
done
