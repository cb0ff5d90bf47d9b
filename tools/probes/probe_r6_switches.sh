#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# round 6's run-time switches, each against the one-rig / few-rig parity and bit-equality tests
R=$GRAFT_REPO_ROOT; cd $R
K="c2_parity or one_launch or chain_handover_soak or batch_matches_single or cholesky_variants or dozen or c1_parity or ragged"
for sw in "PTZ_BA_CHAIN_W0=0" "PTZ_BA_CHAIN_PAIR=0" "PTZ_BA_CHAIN_READY_WHOLE=0" "PTZ_BA_GRAPH_FEW=1" "PTZ_BA_GRAPH_PASSES=3" "PTZ_BA_CHOL_ROWS2=1"; do
  echo "== $sw"; env $sw timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" 2>&1 | tail -1
done
