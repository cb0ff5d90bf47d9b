#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# the chain kernel's tile timeline in the replayed graph (tools/probes/build_tl_lib.sh builds the stamped library), all tiles, full lines
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
PTZCALIB_LIB=$R/tools/probes/hip/lib_chain_tl.so timeout 200 python tools/probes/probe_run.py 1 3 2>&1 | grep "^tl \|lm_steps" | tail -120 > gpurun_out/${OUT:-r6_tl_full.txt}
