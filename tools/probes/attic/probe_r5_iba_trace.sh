#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# PTZ-IBA, 64 rigs in lock step: where a round's time goes (PTZ_BATCHER_TRACE: per view batch create / set / solve / get / destroy)
R=$GRAFT_REPO_ROOT; cd $R
PTZ_BATCHER_TRACE=1 timeout 600 python tools/probes/probe_iba_batch.py 64 200 2> /tmp/iba_trace.txt | grep -E "rigs" | tail -1 | cut -c1-300
python3 - <<'PY'
import re, statistics as st
rows = [list(map(float, re.findall(r"([0-9.]+) ms|create ([0-9.]+) set ([0-9.]+) solve ([0-9.]+) get ([0-9.]+) destroy ([0-9.]+)", l)[0][1:])) for l in open("/tmp/iba_trace.txt") if l.startswith("batcher views")]
n = len(rows) // 3  # warm-up call of 4 rigs, then two runs of 64: take the last third
rows = rows[-128:]
for i, k in enumerate(("create", "set", "solve", "get", "destroy")):
    v = [r[i] for r in rows]; print(f"{k:8s} sum {sum(v):7.1f} ms  mean {st.mean(v):.3f}  median {st.median(v):.3f}  max {max(v):.3f}")
rr = [float(re.findall(r"ran ([0-9.]+) ms", l)[0]) for l in open("/tmp/iba_trace.txt") if l.startswith("batcher round ran")]
rr = rr[-146:]; print("rounds", len(rr), "sum %.1f ms mean %.3f" % (sum(rr), st.mean(rr)))
PY
grep -E "^batcher views" /tmp/iba_trace.txt | tail -130 | awk 'NR%16==1' | cut -c1-150
