#!/bin/bash
# elimination order A/B: dissected (default) against natural, one rig and a 256-scene batch
R=$GRAFT_REPO_ROOT; T=${1:-order}
mkdir -p $R/gpurun_out/$T; cd $R
for o in nd natural; do
  echo "== PTZ_BA_ORDER=$o" | tee -a gpurun_out/$T/timing.txt
  PTZ_BA_ORDER=$o timeout 300 python tests/probe_timing.py 1 256 2>&1 | grep '^{' | tee -a gpurun_out/$T/timing.txt
done
