#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# round 6: correctness subset around the one-rig path, then A/B of one rig's wall time (tools/probes/hip/lib_prev.so = round 5's library)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-chol or c2_parity or one_launch or chain or back_substitution or c1_parity or batch_matches_single or repeated_solves or control_inside or eval_with_four}" 2>&1 | tail -5
IBA=0 bash tools/probes/probe_r5_single_ab.sh 2>&1 | tee gpurun_out/r6_single_ab.txt
