#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# kernel trace of the 64-rig PTZ-IBA run: device-busy time per kernel against the wall time of the calls
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r4_iba_ktrace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PTZ_IBA_COHORTS=${2:-1}
timeout ${KT_TIMEOUT:-200} rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/probes/probe_iba_batch.py 64 200 > $O/log.txt 2>&1
tail -2 $O/log.txt | cut -c1-400
python3 - <<PY
import csv, glob, collections
f = glob.glob('$O/t/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# the last repetition = the kernels after the largest idle gap in the last 60 % of the trace
n = len(rows); lo = int(n * 0.4)
gaps = [(rows[i + 1][0] - rows[i][1], i + 1) for i in range(lo, n - 1)]
g, cut = max(gaps)
last = rows[cut:]
span = (last[-1][1] - last[0][0]) / 1e6
busy = sum(e - s for s, e, _ in last) / 1e6
print('last repetition: %d kernels, span %.1f ms, busy %.1f ms' % (len(last), span, busy))
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, k in last:
    k = k.replace('ptz::(anonymous namespace)::', '').replace('void ', '')[:60]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e6
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:40]: print('%-60s %6d %8.2f ms  %6.1f us' % (k, c, t, 1e3 * t / c))
PY
find $O -name "*kernel_trace.csv" -size +20M -delete
