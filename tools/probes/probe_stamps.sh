#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# where a k_schur workgroup's time goes: a library built with -DPTZ_SCHUR_STAMPS (tools/probes/hip/lib_stamps.so) prints the
# phase durations of one camera's workgroup per launch; one rig alone, then inside a 64-scene batch
R=$GRAFT_REPO_ROOT; T=${1:-stamps}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
export PTZCALIB_LIB=$R/tools/probes/hip/lib_stamps.so
PTZ_BA_GRAPH=0 timeout 200 python tools/probes/probe_run.py 1 1 2>&1 | grep -E "k_schur|lm_steps" | sort | uniq -c | sort -rn | head -12 | tee $O/one.txt
PTZ_BA_GRAPH=0 PTZ_BA_STREAMS=1 timeout 200 python tools/probes/probe_run.py 64 1 2>&1 | grep -E "k_schur|lm_steps" | sort | uniq -c | sort -rn | head -12 | tee $O/b64.txt
unset PTZCALIB_LIB
[ "${TESTS:-0}" = "1" ] && timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q --timeout 600 -k "iba_batch or c4_full or c3" 2>&1 | tail -8 | tee $O/pytest_new.txt
