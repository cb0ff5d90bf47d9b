#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u; cd $GRAFT_REPO_ROOT
# Probe (not a test): HIP API statistics of the lock-step PTZ-IBA batch (which runtime call stalls)
out=gpurun_out/${1:-iba_hiptrace}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats -d /tmp/iba_prof -o iba --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probes/probe_iba_batch.py 64 200 > $GRAFT_REPO_ROOT/$out/iba.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -2 $out/iba.txt
find /tmp/iba_prof -name "*stats*" | head
f=$(find /tmp/iba_prof -name "*hip_api_stats.csv" | head -1)
cp $f $out/hip_api_stats.csv
head -25 $f
t=$(find /tmp/iba_prof -name "*hip_api_trace.csv" | head -1)
ls -la $t
# the calls that took longer than 5 ms
python3 - "$t" <<'PY' > $out/slow_calls.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
slow = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], r["Thread_Id"], int(r["Start_Timestamp"])) for r in rows]
slow = [s for s in slow if s[0] > 5e6]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for d, f, th, st in sorted(slow, key=lambda x: x[3]):
    print("%.1f ms  %-28s thread %s  at %.1f ms" % (d / 1e6, f, th, (st - t0) / 1e6))
PY
tail -60 $out/slow_calls.txt
