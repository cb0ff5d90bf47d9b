#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
R=$GRAFT_REPO_ROOT; T=${1:-r02_prof}; N=${2:-1000}
O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PTZ_BA_STREAMS=1
timeout 400 python3 $R/tools/probes/probe_c4pmc.py $N   # fills the scene cache
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc2 -- python3 $R/tools/probes/probe_c4pmc.py $N > /dev/null 2>&1; echo "pmc2 rc=$?"
timeout 500 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc5 -- python3 $R/tools/probes/probe_c4pmc.py $N > /dev/null 2>&1; echo "pmc5 rc=$?"
timeout 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc4 -- python3 $R/tools/probes/probe_c4pmc.py $N > /dev/null 2>&1; echo "pmc4 rc=$?"
find $O -name "*kernel_trace.csv" -size +30M -delete
cd $R; python3 profiles/summarize_pmc.py $O/pmc2 $O/pmc4 $O/pmc5 > $O/pmc_summary_b.json; head -c 600 $O/pmc_summary_b.json
