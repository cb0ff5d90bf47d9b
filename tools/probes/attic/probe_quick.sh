#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# quick check after a kernel change: the solver-centred GPU tests + single-rig timing (+ optional batch timing)
R=$GRAFT_REPO_ROOT; T=${1:-quick}; shift || true
mkdir -p $R/gpurun_out/$T; cd $R
timeout ${PT:-300} python -m pytest tests -m gpu -x -q --timeout 120 -k "${K:-chol or ba_ or trajector or linearize or cpp_ptzray}" 2>&1 | tail -6 | tee gpurun_out/$T/pytest.txt
timeout 200 python tools/probes/probe_timing.py ${@:-1} 2>&1 | tail -3 | tee gpurun_out/$T/timing.txt
