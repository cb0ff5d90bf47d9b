#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# PTZ-IBA, 64 rigs in lock step: tools/probes/hip/lib_prev.so against the working tree's library, alternating
R=$GRAFT_REPO_ROOT; cd $R
[ "${TESTS:-1}" = "1" ] && timeout 1200 python -m pytest tests -x -q -m gpu -k "${K:-view or iba or rig or worldcup or incremental or batcher}" 2>&1 | tail -4
for l in prev product prev product prev product; do
  if [ $l = product ]; then unset PTZCALIB_LIB; else export PTZCALIB_LIB=$R/tools/probes/hip/lib_$l.so; fi
  echo "== iba $l"; timeout 600 python tools/probes/probe_iba_batch.py 64 200 2>&1 | grep -E "rigs" | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: round(d[k], 1) if isinstance(d[k], float) else d[k] for k in ('wall_total_ms', 'wall_ms', 'views_per_s', 'ba_ms', 'krt_ms')})"
done
