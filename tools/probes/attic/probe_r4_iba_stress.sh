#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# the 64-rig lock step N times over (each run: warm-up + two timed repetitions, results checked against the solo run): failures, hangs
R=$GRAFT_REPO_ROOT; cd $R; N=${1:-10}
ok=0; bad=0
for i in $(seq 1 $N); do
  if timeout 120 python tools/probes/probe_iba_batch.py 64 200 > /tmp/stress_$i.txt 2>&1; then ok=$((ok+1)); tail -1 /tmp/stress_$i.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['wall_ms']), end=' ')"
  else bad=$((bad+1)); echo; echo "run $i FAILED rc=$?"; tail -5 /tmp/stress_$i.txt; fi
done
echo; echo "ok $ok failed $bad"
