#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# one GPU call per development step: the whole GPU suite, the k_schur phase stamps, new kernels against round 2's (A/B), timing
R=$GRAFT_REPO_ROOT; T=${1:-round}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -12 | tee $O/pytest.txt
PTZCALIB_LIB=$R/tools/probes/hip/lib_stamps.so PTZ_BA_GRAPH=0 PTZ_BA_STREAMS=1 timeout 200 python tools/probes/probe_run.py 64 1 2>&1 | grep -E "k_schur" | sort | uniq -c | sort -rn | sed -n '3,8p' | tee $O/stamps.txt
for rep in 1 2; do for w in 0 1; do
  echo "== PTZ_BA_SCHUR_W=$w" | tee -a $O/timing.txt
  PTZ_BA_SCHUR_W=$w timeout 300 python tools/probes/probe_timing.py ${SIZES:-1 256} 2>&1 | grep '^{' | tee -a $O/timing.txt
done; done
if [ "${KRT:-1}" = "1" ]; then for g in 64 16; do
  echo "== PTZ_KRT_GROUP=$g" | tee -a $O/krt.txt
  PTZ_KRT_GROUP=$g timeout 200 python tools/probes/probe_krt.py 100000 0 2>&1 | grep '^{' | tee -a $O/krt.txt
  PTZ_KRT_GROUP=$g timeout 200 python tools/probes/probe_krt.py 100000 1 2>&1 | grep '^{' | head -1 | tee -a $O/krt.txt
done; fi
[ "${OVH:-0}" = "1" ] && timeout 600 python tools/probes/probe_batch_overheads.py 64 2>&1 | tail -12 | tee $O/overheads.txt
[ "${IBA:-0}" = "1" ] && timeout 900 python tools/probes/probe_iba_batch.py 64 200 2>&1 | tail -6 | tee $O/iba.txt
