#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# per-scene accuracy of the C4 batch with the current Schur kernel and with round 2's (PTZ_BA_SCHUR_W=1)
R=$GRAFT_REPO_ROOT; cd $R
N=${1:-1000}
PTZ_BA_SCHUR_W=0 timeout 600 python tools/probes/probe_c4_check.py w0 $N 2>&1 | tail -25
PTZ_BA_SCHUR_W=1 timeout 600 python tools/probes/probe_c4_check.py w1 $N 2>&1 | tail -25
python3 - <<PY
import json
a=json.load(open("gpurun_out/c4check/w0.json")); b=json.load(open("gpurun_out/c4check/w1.json"))
d=[(x["i"], x["it"], y["it"], x["cost"], y["cost"], x["ferr"], y["ferr"]) for x,y in zip(a,b) if x["it"]!=y["it"] or abs(x["cost"]-y["cost"])>1e-9*abs(y["cost"])]
print("scenes whose iteration count or cost differs between the kernels:", len(d))
for r in d[:30]: print("  ", r)
PY
