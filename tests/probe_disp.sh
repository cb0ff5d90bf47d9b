#!/bin/bash
# PTZRayDistDisp bring-up: its GPU tests first, then the solver-centred tests of the other types
R=$GRAFT_REPO_ROOT; T=${1:-disp}
mkdir -p $R/gpurun_out/$T; cd $R
timeout ${PT:-400} python -m pytest tests/test_gpu_disp.py -m gpu -q --timeout 150 2>&1 | tail -40 | tee gpurun_out/$T/disp.txt
timeout 400 python -m pytest tests -m gpu -x -q --timeout 120 -k "chol or ba_ or trajector or linearize or cpp_ptzray" --deselect tests/test_gpu_disp.py 2>&1 | tail -6 | tee gpurun_out/$T/pytest.txt
timeout 200 python tests/probe_timing.py 1 2>&1 | tail -3 | tee gpurun_out/$T/timing.txt
