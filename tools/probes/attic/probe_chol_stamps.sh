#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# time stamps inside chol_col_step's critical workgroup (one rig), for the probe builds given as arguments
# (tools/probes/hip/lib_<name>.so, built with -DPTZ_CHOL_STAMPS); prints the median line per (step, tile)
R=$GRAFT_REPO_ROOT; cd $R
for l in "$@"; do
  echo "== $l"
  PTZCALIB_LIB=$R/tools/probes/hip/lib_$l.so PTZ_BA_GRAPH=0 timeout 200 python tools/probes/probe_run.py 1 1 2>&1 | grep "^chol_col_step" | python3 -c "
import sys, re, statistics
rows = {}
for l in sys.stdin:
    m = re.match(r'chol_col_step (\d+) tile (\d+) updates (\d+) \| x10 ns: prologue (\d+), first operands \(load \+ solve\) (\d+), C \+ first update (\d+), further columns (\d+), to LDS (\d+), diagonal factor (\d+)', l)
    if m: rows.setdefault(tuple(map(int, m.groups()[:3])), []).append(list(map(int, m.groups()[3:])))
for k in sorted(rows):
    v = rows[k]; med = [statistics.median(c) / 100.0 for c in zip(*v)]
    print('step %d tile %d updates %d (n=%d): prologue %.1f operands %.1f update %.1f further %.1f toLDS %.1f diag %.1f us' % (*k, len(v), *med))
"
done
# the one-launch factorisation (chol_chain_kernel): the diagonal tiles of one pass in the middle of the solve, raw
if [ -n "${CHAIN:-}" ]; then
  PTZCALIB_LIB=$R/tools/probes/hip/lib_$CHAIN.so PTZ_BA_GRAPH=0 timeout 200 python tools/probes/probe_run.py 1 1 2>&1 | grep "^chol_chain" | sed -n '131,156p'
fi
