#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# Round 4: PTZ-IBA, 64 rigs x 200 views in lock step, for several cohort counts / batch splits, views of resident tracks vs host packing
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do
  echo "== [$v]"
  ( for kv in $v; do export "$kv"; done; timeout 600 python tools/probes/probe_iba_batch.py 64 200 2>&1 | grep -E "solo|rigs" | tail -3 )
done
