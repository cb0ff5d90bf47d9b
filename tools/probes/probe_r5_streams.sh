#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# the library's default pipeline on the C4 batch by number of scene groups (PTZ_BA_STREAMS) and run-ahead (PTZ_BA_AHEAD)
R=$GRAFT_REPO_ROOT; cd $R
for s in ${STREAMS:-1 2 3 4}; do
  echo "== PTZ_BA_STREAMS=$s"; PTZ_BA_STREAMS=$s timeout 600 python tools/probes/probe_c4.py 1000 2>&1 | grep -E "^solve|lm steps" | head -4
done
