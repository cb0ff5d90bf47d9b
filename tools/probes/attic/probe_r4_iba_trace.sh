#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
R=$GRAFT_REPO_ROOT; cd $R
export PTZ_IBA_COHORTS=${1:-1} PTZ_BATCHER_TRACE=1
timeout 600 python tools/probes/probe_iba_batch.py 64 200 > /tmp/iba_trace.txt 2>&1
python3 - <<'PY'
import re
v = [l for l in open('/tmp/iba_trace.txt') if l.startswith('batcher views')]
k = [l for l in open('/tmp/iba_trace.txt') if l.startswith('batcher krt')]
half = len(v) // 2
v = v[half:]; k = k[len(k) // 2:]   # the second (timed) repetition
tot = dict(create=0, set=0, solve=0, get=0, destroy=0); n = 0; cams = 0
for l in v:
    m = re.search(r'n=(\d+) cams=(\d+) create ([\d.]+) set ([\d.]+) solve ([\d.]+) get ([\d.]+) destroy ([\d.]+)', l)
    n += int(m.group(1)); cams += int(m.group(2))
    for key, g in zip(tot, m.groups()[2:]): tot[key] += float(g)
print("view batches", len(v), "problems", n, "cams", cams, {a: round(b, 1) for a, b in tot.items()}, "sum", round(sum(tot.values()), 1))
pk = ck = dk = 0
for l in k:
    m = re.search(r'pack ([\d.]+) call ([\d.]+) device ([\d.]+)', l)
    pk += float(m.group(1)); ck += float(m.group(2)); dk += float(m.group(3))
print("krt launches", len(k), "pack %.1f call %.1f device %.1f ms" % (pk, ck, dk))
print(open('/tmp/iba_trace.txt').read().splitlines()[-1][:400])
PY
