#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# PTZ-IBA, 64 rigs: in-call time against the number of independent lock steps (PTZ_IBA_COHORTS), two alternations
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for k in ${KS:-1 2 3 4}; do
  echo "== cohorts $k"; PTZ_IBA_COHORTS=$k timeout 600 python tools/probes/probe_iba_batch.py 64 200 2>&1 | grep -E "rigs" | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: round(d[k], 1) if isinstance(d[k], float) else d[k] for k in ('wall_total_ms', 'wall_ms', 'views_per_s', 'rounds', 'ba_batches', 'ba_ms', 'krt_ms')})"
done; done
