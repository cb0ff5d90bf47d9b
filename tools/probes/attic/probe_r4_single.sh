#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# Round 4, single rig (C2): per-kernel durations of graph-replayed one-rig solves (rocprofv3 kernel trace), for the default
# library and for variants selected by environment switches.  usage: tools/probes/probe_r4_single.sh <tag> ["VAR=val ..." ...]
R=$GRAFT_REPO_ROOT; T=${1:-r4single}; shift || true
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
run() {  # $1 = label, $2 = "VAR=val VAR=val" (exported for this run only)
  ( for kv in $2; do export "$kv"; done
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/$1 -- python3 $R/tools/probes/probe_run.py 1 6 > $R/gpurun_out/$T/$1.log 2>&1 )
  find $R/gpurun_out/$T/$1 -name "*kernel_trace.csv" -size +60M -delete
  echo "== $1 [$2]"; tail -1 $R/gpurun_out/$T/$1.log; python3 $R/tools/kstats.py $R/gpurun_out/$T/$1 | sort -t't' -k3 | head -24
}
run default ""
i=0
for v in "$@"; do i=$((i+1)); run "var$i" "$v"; done
