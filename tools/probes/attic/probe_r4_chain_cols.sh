#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
R=$GRAFT_REPO_ROOT; cd $R
PTZCALIB_LIB=$R/tools/probes/hip/lib_$1.so PTZ_BA_GRAPH=0 timeout 200 python tools/probes/probe_run.py 1 1 2>&1 | grep "^chol_chain" > /tmp/chain_all.txt
python3 - <<'PY'
import re
lines = open('/tmp/chain_all.txt').read().splitlines()
# passes: split at tile 12's summary line
passes, cur = [], []
for l in lines:
    cur.append(l)
    if l.startswith('chol_chain tile 12 '):
        pass
for l in lines:
    pass
# take the 11th pass: count 'chol_chain tile 12 updates' occurrences
idx = [i for i, l in enumerate(lines) if l.startswith('chol_chain tile 12 ')]
if len(idx) > 11:
    # lines of pass 11: between the first 'tile' line after idx[9]'s column lines and idx[10]'s column lines
    start = idx[9]; end = idx[10]
    seg = lines[start:end + 14]
else:
    seg = lines
post = {}
rows = []
for l in seg:
    m = re.match(r'chol_chain tile (\d+) updates (\d+) .*seen (\d+) (\d+) (\d+) (\d+), solved\+updated (\d+), in LDS (\d+), factored (\d+), posted (\d+)', l)
    if m: rows.append(('t',) + tuple(map(int, m.groups())))
    m = re.match(r'chol_chaincol tile (\d+) column (\d+): T seen (\d+), block 0 seen (\d+), applied (\d+)', l)
    if m: rows.append(('c',) + tuple(map(int, m.groups())))
ts = [r[10] for r in rows if r[0] == 't']
base = max(ts) - 30000 if ts else 0
for r in rows:
    if r[0] == 't' and r[10] >= base - 5000:
        print('tile %2d upd %2d: F %s solved %.2f inLDS %.2f factored %.2f posted %.2f' % (r[1], r[2], ' '.join('%.2f' % ((x - base) / 100.0) for x in r[3:7]), *[(x - base) / 100.0 for x in r[7:11]]))
    if r[0] == 'c' and r[5] >= base - 5000:
        print('      tile %2d col %2d: T %.2f  F0 %.2f  applied %.2f' % (r[1], r[2], *[(x - base) / 100.0 for x in r[3:6]]))
PY
