#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# stamps of one k_schur workgroup inside a 64-scene batch + the lock-step PTZ-IBA probe
R=$GRAFT_REPO_ROOT; T=${1:-quick2}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
PTZCALIB_LIB=$R/tools/probes/hip/lib_stamps.so PTZ_BA_GRAPH=0 PTZ_BA_STREAMS=1 timeout 200 python tools/probes/probe_run.py 64 1 2>&1 | grep -E "k_schur" | sort | uniq -c | sort -rn | sed -n '3,8p' | tee $O/stamps.txt
timeout 900 python tools/probes/probe_iba_batch.py ${RIGS:-64} 200 2>&1 | tail -6 | tee $O/iba.txt
