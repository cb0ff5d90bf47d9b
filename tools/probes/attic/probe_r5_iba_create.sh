#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# PTZ-IBA, 64 rigs in lock step: where ptz_ba_batch_create_views spends its time (PTZ_BA_DEBUG_TIMING lines of the last run)
R=$GRAFT_REPO_ROOT; cd $R
PTZ_BA_DEBUG_TIMING=1 timeout 600 python tools/probes/probe_iba_batch.py 64 200 2> /tmp/iba_dbg.txt | grep -E "rigs" | tail -1 | cut -c1-200
python3 - <<'PY'
import re, statistics as st
rows = []
for l in open("/tmp/iba_dbg.txt"):
    m = re.search(r"\[ptz_ba_create\] host structure ([0-9.]+) ms \(observations ([0-9.]+), pair entries ([0-9.]+)\), uploads \+ allocations ([0-9.]+) ms, mask \+ rest ([0-9.]+) ms", l)
    if m: rows.append(list(map(float, m.groups())))
rows = rows[-128:]
for i, k in enumerate(("host structure", "observations", "pair entries", "uploads + allocations", "mask + rest")):
    v = [r[i] for r in rows]; print(f"{k:24s} sum {sum(v):7.1f} ms  mean {st.mean(v):.3f}  max {max(v):.3f}")
PY
grep -v "ptz_ba_create\] host\|launch shape\|group " /tmp/iba_dbg.txt | cut -c1-160 | sort | uniq -c | sort -rn | head -12
python3 - <<'PY'
import re, statistics as st
rows = []
for l in open("/tmp/iba_dbg.txt"):
    m = re.search(r"\[ptz_ba\] groups (\d+): total ([0-9.]+) ms, enqueue ([0-9.]+) ms, sync-wait ([0-9.]+) ms, device ([0-9.]+) ms", l)
    if m: rows.append(list(map(float, m.groups())))
rows = rows[-128:]
for i, k in enumerate(("groups", "total", "enqueue", "sync-wait", "device")):
    v = [r[i] for r in rows]; print(f"solve {k:12s} sum {sum(v):7.1f} ms  mean {st.mean(v):.3f}  max {max(v):.3f}")
PY
grep "passes enqueued" /tmp/iba_dbg.txt | tail -128 | sed 's/.*: \([0-9]*\) passes enqueued, \([0-9]*\) reached.*/\1 \2/' | sort | uniq -c | sort -rn | head -8
python3 - <<'PY'
import re, statistics as st
rows = []
for l in open("/tmp/iba_dbg.txt"):
    m = re.search(r"views: allocations \+ upload ([0-9.]+) ms, build enqueued in ([0-9.]+) ms, waited ([0-9.]+) ms", l)
    if m: rows.append(list(map(float, m.groups())))
rows = rows[-128:]
for i, k in enumerate(("allocations + upload", "build enqueued", "waited for the build")):
    v = [r[i] for r in rows]; print(f"views {k:24s} sum {sum(v):7.1f} ms  mean {st.mean(v):.3f}  max {max(v):.3f}")
PY
