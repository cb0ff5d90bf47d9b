#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# the solver parity suite under the non-default settings of the library's run-time switches: every switch is
# documented as "same results", so the suite must pass whichever way it is set
R=$GRAFT_REPO_ROOT; cd $R
for e in "PTZ_BA_CHOL_CHAIN=0" "PTZ_BA_CHOL_CHAIN_MAX=8" "PTZ_BA_CHOL_HALFK=0" "PTZ_BA_GPU_STRUCT=0" "PTZ_BA_GRAPH=0" "PTZ_BA_BACKSOLVE_HOST_LIST=0" "PTZ_BA_COMPACT=0" "PTZ_BA_STREAMS=3"; do  # (PTZ_BA_CHOL_FUSED=0 agrees to round-off only: right-looking updates in block-column order)
  echo "== $e"; env $e timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_disp.py -m gpu -x -q 2>&1 | tail -2
done
