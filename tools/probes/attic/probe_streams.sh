#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# scene groups (PTZ_BA_STREAMS) at C4 size: library-default pipeline time per solve
R=$GRAFT_REPO_ROOT; cd $R
timeout 400 python3 tools/probes/probe_c4pmc.py 1000 > /dev/null 2>&1   # scene cache
for g in 1 2 3 4 6; do echo "streams $g: $(PTZ_BA_STREAMS=$g timeout 300 python3 tools/probes/probe_c4pmc.py 1000 2>&1 | tail -1)"; done
