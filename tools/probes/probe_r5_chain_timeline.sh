#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# Where every tile of the C2 rig's one-launch factorisation is when, in the REPLAYED graph: a library built with -DPTZ_CHOL_TIMELINE
# (tools/probes/hip/lib_chain_tl.so: ptz_chol.hip and ptz_ba.hip with that define) stores a few wall-clock stamps per tile and prints
# them from a kernel of its own when the batch is destroyed.  us from the first workgroup's start.
R=$GRAFT_REPO_ROOT; cd $R
PTZCALIB_LIB=$R/tools/probes/hip/lib_chain_tl.so timeout 200 python tools/probes/probe_run.py 1 3 2>&1 | grep "^tl tile\|lm_steps" | tail -95 > /tmp/tl.txt
grep lm_steps /tmp/tl.txt
grep "tl tile ( *\([0-9]*\), *\1)" /tmp/tl.txt | cut -c1-400
echo "--- off-diagonal tiles of the dense tail"
grep "tl tile (\( 9\|10\|11\|12\), *\( 8\| 9\|10\|11\))" /tmp/tl.txt | cut -c1-100
