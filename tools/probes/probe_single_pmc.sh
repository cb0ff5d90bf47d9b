#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# counters of the one-rig solve's kernels (separate --pmc passes, kernel trace only beside them): wave cycles / waits / issue and the
# matrix cores' share, summarised like the C4 passes.  usage: tools/probes/probe_single_pmc.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-single_pmc}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && timeout 200 python tools/probes/probe_run.py 1 1 > /dev/null 2>&1   # (warms the scene cache before any profiler preload)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc3 -- python3 $R/tools/probes/probe_run.py 1 3 > /dev/null 2>&1; echo "pmc3 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc4 -- python3 $R/tools/probes/probe_run.py 1 3 > /dev/null 2>&1; echo "pmc4 rc=$?"
find $O -name "*kernel_trace.csv" -size +30M -delete
cd $R && python3 profiles/summarize_pmc.py $O/pmc3 $O/pmc4 > $O/pmc_summary.json; python3 -c "
import json
d = json.load(open('$O/pmc_summary.json'))
for k, v in d.items():
    if any(x in k for x in ('chain', 'backsolve', 'k_eval', 'k_schur')): print(k[:50], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})"
