#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# Round artefacts for profiles/: bench line under the kernel trace, then separate PMC passes on the same C4 workload.
# usage: tools/probes/probe_profile2.sh <tag> [scenes]
R=$GRAFT_REPO_ROOT; T=${1:-r02f}; N=${2:-1000}
O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PTZ_BA_STREAMS=1   # one scene group for everything profiled: every launch covers the whole batch, as in bench.py's timed region
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --headline-only --scenes $N --steps 2 > $O/bench_under_rocprof.json 2>$O/bench_under_rocprof.err; echo "stats rc=$?"
timeout 400 python3 $R/tools/probes/probe_c4pmc.py $N   # fills the scene cache
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc1 -- python3 $R/tools/probes/probe_c4pmc.py $N > /dev/null 2>&1; echo "pmc1 rc=$?"
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc2 -- python3 $R/tools/probes/probe_c4pmc.py $N > /dev/null 2>&1; echo "pmc2 rc=$?"
timeout 500 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc5 -- python3 $R/tools/probes/probe_c4pmc.py $N > /dev/null 2>&1; echo "pmc5 rc=$?"
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc3 -- python3 $R/tools/probes/probe_c4pmc.py $N > /dev/null 2>&1; echo "pmc3 rc=$?"
timeout 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc4 -- python3 $R/tools/probes/probe_c4pmc.py $N > /dev/null 2>&1; echo "pmc4 rc=$?"
find $O -name "*kernel_trace.csv" -size +30M -delete
cd $R
python3 profiles/summarize_pmc.py $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 > $O/pmc_summary.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
ls $O; tail -c 300 $O/bench_under_rocprof.json; python3 tools/kstats.py $O/kernel_stats.csv | head -20
