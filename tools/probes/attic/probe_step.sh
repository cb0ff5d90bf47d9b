#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# after a kernel change: the solver parity suites, then device time per kernel family for one rig, 8 rigs and 256 scenes
# usage: tools/probes/probe_step.sh <tag> [pytest -k expression]
R=$GRAFT_REPO_ROOT; T=${1:-step}; K=${2:-}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -8 | tee $O/pytest.txt
else timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_disp.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt; fi
timeout 600 python tools/probes/probe_timing.py 1 8 256 2>$O/timing.err | tee $O/timing.json
if [ "${CHAIN_AB:-0}" = "1" ]; then
  for v in 0 1 0 1; do echo "== PTZ_BA_CHOL_CHAIN=$v"; PTZ_BA_CHOL_CHAIN_MAX=8 PTZ_BA_CHOL_CHAIN=$v timeout 300 python tools/probes/probe_timing.py ${CHAIN_SIZES:-1 2 3 4} 2>>$O/timing.err | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['B'], d['dev_ms'], round(d['it_per_s']), d['lm_steps'], d['profile_ms'])"; done | tee $O/chain_ab.txt
fi
