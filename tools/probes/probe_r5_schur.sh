#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# round 5: k_schur_f (factored 8-double rows, three workgroups per compute unit) against k_schur (PTZ_BA_SCHUR_F=0) on ONE box:
# the BA parity tests, then 256 scenes (4 seeds cycled) and the C4 batch, per-family device times.
# usage: probe_r5_schur.sh <tag>      TESTS=0 skips pytest, C4=0 the 1000-scene batch
R=$GRAFT_REPO_ROOT; T=${1:-r5_schur}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 900 -k "${K:-ba_ or linearize or schur or batch}" 2>&1 | tail -15 | tee $O/pytest.txt
fi
for rep in 1 2; do for f in 0 1; do
  echo "== PTZ_BA_SCHUR_F=$f" | tee -a $O/timing.txt
  PTZ_BA_SCHUR_F=$f timeout 300 python tools/probes/probe_timing.py ${SIZES:-256} 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['B'], d['dev_ms'], round(d['it_per_s']), d['profile_ms'])" | tee -a $O/timing.txt
done; done
if [ "${C4:-1}" = "1" ]; then for f in 0 1 0 1; do
  echo "== C4 PTZ_BA_SCHUR_F=$f" | tee -a $O/timing.txt
  PTZ_BA_SCHUR_F=$f PTZ_BA_STREAMS=1 timeout 600 python tools/probes/probe_c4_families.py 1000 2>&1 | grep '^{' | cut -c1-700 | tee -a $O/timing.txt
done; fi
if [ "${PMC:-0}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  export PTZ_BA_STREAMS=1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc3 -- python3 $R/tools/probes/probe_run.py 256 1 > /dev/null 2>&1; echo "pmc3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/pmc6 -- python3 $R/tools/probes/probe_run.py 256 1 > /dev/null 2>&1; echo "pmc6 rc=$?"
  find $O -name "*kernel_trace.csv" -size +30M -delete
  cd $R; python3 profiles/summarize_pmc.py $O/pmc3 $O/pmc6 > $O/pmc_summary.json
  python3 - <<PY
import json
d=json.load(open("$O/pmc_summary.json"))
for k in d:
    if "schur" in k: print(k, json.dumps(d[k]))
PY
  find $O -name "*counter_collection.csv" -size +20M -delete
fi
