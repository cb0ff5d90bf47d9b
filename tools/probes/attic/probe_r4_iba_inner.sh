#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# where the view batches of a 64-rig PTZ-IBA run spend their time INSIDE the library (PTZ_BA_DEBUG_TIMING lines, second repetition)
R=$GRAFT_REPO_ROOT; cd $R
export PTZ_IBA_COHORTS=${1:-1} PTZ_BATCHER_TRACE=1 PTZ_BA_DEBUG_TIMING=1
timeout 600 python tools/probes/probe_iba_batch.py 64 200 > /tmp/iba_trace.txt 2>&1
python3 - <<'PY'
import re
L = open('/tmp/iba_trace.txt').read().splitlines()
def second_half(p): v = [l for l in L if l.startswith(p)]; return v[len(v) // 2:]
c = second_half('[ptz_ba_create] host structure')
tot = [0.0] * 5
for l in c:
    m = re.search(r'host structure ([\d.]+) ms \(observations ([\d.]+), pair entries ([\d.]+)\), uploads \+ allocations ([\d.]+) ms, mask \+ rest ([\d.]+)', l)
    for i in range(5): tot[i] += float(m.group(i + 1))
print('creates', len(c), 'host structure %.1f (obs %.1f, entries %.1f) uploads+allocs %.1f mask+rest %.1f ms' % tuple(tot))
g = second_half('[ptz_ba] groups')
tot = [0.0] * 4
for l in g:
    m = re.search(r'total ([\d.]+) ms, enqueue ([\d.]+) ms, sync-wait ([\d.]+) ms, device ([\d.]+)', l)
    for i in range(4): tot[i] += float(m.group(i + 1))
print('solves', len(g), 'total %.1f enqueue %.1f sync-wait %.1f device %.1f ms' % tuple(tot))
for p in ('[ptz_ba_create] view', '[ptz_ba_create] stage'):
    v = second_half(p)
    if v: print(len(v), v[-1][:300])
v = second_half('batcher views'); print(len(v)); print('\n'.join(v[-3:]))
print(L[-1][:400])
PY
