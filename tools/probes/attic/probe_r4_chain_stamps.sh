#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# absolute time stamps of the diagonal workgroups of chol_chain_kernel (one rig, one pass in the middle of the solve)
# usage: tools/probes/probe_r4_chain_stamps.sh <lib name under tools/probes/hip (lib_<name>.so)>
R=$GRAFT_REPO_ROOT; cd $R
PTZCALIB_LIB=$R/tools/probes/hip/lib_$1.so PTZ_BA_GRAPH=0 timeout 200 python tools/probes/probe_run.py 1 1 2>&1 | grep "^chol_chain" | sed -n '131,143p' | python3 -c "
import sys, re
rows = []
for l in sys.stdin:
    m = re.match(r'chol_chain tile (\d+) updates (\d+) .*seen (\d+) (\d+) (\d+) (\d+), solved\+updated (\d+), in LDS (\d+), factored (\d+), posted (\d+)', l)
    if m: rows.append(list(map(int, m.groups())))
if rows:
    base = min(r[9] for r in rows)
    for r in sorted(rows, key=lambda r: r[9]):
        f = [(x - base) / 100.0 if x else float('nan') for x in r[2:]]
        print('tile %2d updates %d | F0 %7.2f F1 %7.2f F2 %7.2f F3 %7.2f | solved %7.2f inLDS %7.2f factored %7.2f posted %7.2f us' % (r[0], r[1], *f))
"
