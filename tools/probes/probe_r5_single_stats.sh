#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# one rig (C2): kernel statistics of the product library (or PTZCALIB_LIB), rocprofv3 --kernel-trace --stats
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r5_single_stats}; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/single -- python3 $R/tools/probes/probe_run.py 1 6 > $O/single.log 2>&1; echo "single rc=$?"
cp $(find $O/single -name "*kernel_stats.csv" | head -1) $O/single_rig_kernel_stats.csv
tail -1 $O/single.log; python3 $R/tools/kstats.py $O/single_rig_kernel_stats.csv | head -14
