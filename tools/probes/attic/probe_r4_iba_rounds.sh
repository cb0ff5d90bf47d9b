#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# a 64-rig PTZ-IBA run, per lock-step round: the clients' own (host) work between rounds against the rounds' device calls
R=$GRAFT_REPO_ROOT; cd $R
echo "host cores: $(nproc)"; lscpu | grep -E "Model name|^CPU\(s\)|Thread" | head -4
export PTZ_IBA_COHORTS=${1:-1} PTZ_BATCHER_TRACE=1
timeout 600 python tools/probes/probe_iba_batch.py 64 200 > /tmp/iba_trace.txt 2>&1
python3 - <<'PY'
import re
L = open('/tmp/iba_trace.txt').read().splitlines()
def second_half(p): v = [l for l in L if l.startswith(p)]; return v[len(v) // 2:]
th = [float(re.search(r'last round ([\d.]+) ms', l).group(1)) for l in second_half('batcher round clients')]
rn = [float(re.search(r'ran ([\d.]+) ms', l).group(1)) for l in second_half('batcher round ran')]
print('rounds %d: clients own work %.1f ms (median %.2f, max %.2f), rounds ran %.1f ms (median %.2f)' % (len(th), sum(th), sorted(th)[len(th)//2], max(th), sum(rn), sorted(rn)[len(rn)//2]))
print(L[-1][:400])
PY
