#!/bin/bash
# pass pipeline variants: graph replay on/off, run-ahead depth; single rig and a 64-scene batch
R=$GRAFT_REPO_ROOT; T=${1:-loop}
mkdir -p $R/gpurun_out/$T
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/$T/pytest.txt
for cfg in "1 3" "0 3" "1 2" "1 6"; do
  set -- $cfg
  echo "== PTZ_BA_GRAPH=$1 PTZ_BA_AHEAD=$2" >> gpurun_out/$T/timing.txt
  PTZ_BA_GRAPH=$1 PTZ_BA_AHEAD=$2 timeout 300 python tests/probe_timing.py 1 64 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['B'], 'wall', [round(x*1e3,2) for x in d['wall_s']], 'dev', [round(x,2) for x in d['dev_ms']], 'steps', d['lm_steps'], 'it/s', round(d['it_per_s']))
" >> gpurun_out/$T/timing.txt
done
cat gpurun_out/$T/pytest.txt gpurun_out/$T/timing.txt
