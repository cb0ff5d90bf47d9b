#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# Single-rig (C2) critical path: per-kernel durations of a one-scene solve from a rocprofv3 kernel trace,
# plus the in-library per-family event timing.  usage: tools/probes/probe_single.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-single}
mkdir -p $R/gpurun_out/$T
cd $R && timeout 300 python tools/probes/probe_timing.py 1 > gpurun_out/$T/timing.json 2> gpurun_out/$T/timing.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats -- python3 $R/tools/probes/probe_run.py 1 3 > $R/gpurun_out/$T/run.log 2>&1
find $R/gpurun_out/$T -name "*kernel_trace.csv" -size +60M -delete
ls -R $R/gpurun_out/$T | head; cat $R/gpurun_out/$T/timing.json; tail -3 $R/gpurun_out/$T/run.log
