#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# elimination orders on one box: nested dissection (default), flat (two lanes), natural; one rig, 8 rigs, 256, the incremental pipeline
R=$GRAFT_REPO_ROOT; T=${1:-order2}
mkdir -p $R/gpurun_out/$T; cd $R
for rep in 1 2; do for o in nd flat natural; do
  echo "== $o" | tee -a gpurun_out/$T/timing.txt
  PTZ_BA_ORDER=$o timeout 300 python tools/probes/probe_timing.py 1 8 256 2>&1 | grep '^{' | tee -a gpurun_out/$T/timing.txt
  PTZ_BA_ORDER=$o timeout 300 python tools/probes/probe_iba.py 2>&1 | tail -2 | tee -a gpurun_out/$T/timing.txt
done; done
