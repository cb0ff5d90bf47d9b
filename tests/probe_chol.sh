#!/bin/bash
R=$GRAFT_REPO_ROOT; T=${1:-chol}
mkdir -p $R/gpurun_out/$T; cd $R
timeout 300 python tests/probe_chol.py 2>&1 | tee gpurun_out/$T/chol.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "chol or ba_c1 or ba_c2 or medium or trajector or bit_identical or large_batch" 2>&1 | tail -8 | tee gpurun_out/$T/pytest.txt
timeout 300 python tests/probe_timing.py 1 2>&1 | tail -1 | tee gpurun_out/$T/timing.txt
