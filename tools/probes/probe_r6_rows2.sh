#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# round 6: chol_update_col_h2 (two row tiles per workgroup) against one (PTZ_BA_CHOL_ROWS2=0), per-family device times of a profiled 256-scene solve
R=$GRAFT_REPO_ROOT; cd $R
[ "${TESTS:-1}" = "1" ] && timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${K:-chol or cholesky_variants or batch_matches_single or medium or ragged or c3_standin or many_cameras}" 2>&1 | tail -4
for r in 0 1 0 1; do
  echo "== rows2=$r"; PTZ_BA_CHOL_ROWS2=$r timeout 300 python tools/probes/probe_timing.py ${SIZES:-256} 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['B'], [round(x,2) for x in d['dev_ms']], round(d['it_per_s']), d['profile_ms'])"
done
