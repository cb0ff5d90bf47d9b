#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# round 3: GPU suite, then the Schur kernel without materialised W (default) against round 2's kernels (PTZ_BA_SCHUR_W=1) on
# ONE box: one rig, 256 scenes (4 seeds cycled), per-family device times; then occupancy / LDS counters of the 256-scene solve.
# usage: probe_schur_ab.sh <tag> [sizes...]      PMC=0 skips the counter pass, AB=0 the legacy kernels
R=$GRAFT_REPO_ROOT; T=${1:-schur_ab}; shift || true
O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -15 | tee $O/pytest.txt
for rep in 1 2; do for w in 0 1; do
  [ "$w" = "0" ] || [ "${AB:-1}" = "1" ] || continue
  echo "== PTZ_BA_SCHUR_W=$w" | tee -a $O/timing.txt
  PTZ_BA_SCHUR_W=$w timeout 300 python tools/probes/probe_timing.py ${@:-1 256} 2>&1 | grep '^{' | tee -a $O/timing.txt
done; done
if [ "${PMC:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  export PTZ_BA_STREAMS=1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc3 -- python3 $R/tools/probes/probe_run.py 256 1 > /dev/null 2>&1; echo "pmc3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/pmc6 -- python3 $R/tools/probes/probe_run.py 256 1 > /dev/null 2>&1; echo "pmc6 rc=$?"
  find $O -name "*kernel_trace.csv" -size +30M -delete
  cd $R; python3 profiles/summarize_pmc.py $O/pmc3 $O/pmc6 > $O/pmc_summary.json
  python3 - <<PY
import json
d=json.load(open("$O/pmc_summary.json"))
for k in ("k_schur","k_lin_cam","k_lin_ray","k_eval","k_ray_prep","chol_update_col"):
    if k in d: print(k, json.dumps(d[k]))
PY
  find $O -name "*counter_collection.csv" -size +20M -delete
fi
