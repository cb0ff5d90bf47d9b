#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# Round-5 evidence for profiles/: the default bench (what the driver runs), the kernel trace of the headline leg and of graph-replayed
# one-rig solves, then the PMC passes of the C4 workload (separate runs per counter group, kernel trace only) and the traffic table.
# usage: tools/probes/probe_r5_evidence.sh <tag> [nobench] [nopmc]
R=$GRAFT_REPO_ROOT; T=${1:-r5_evidence}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
if [ "${2:-}" != "nobench" ]; then
  timeout 1800 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  tail -c 400 $O/bench.err
fi
cd /tmp && export TMPDIR=/tmp
PTZ_BA_STREAMS=1 timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --headline-only --steps 2 > $O/bench_under_rocprof.json 2>$O/bench_under_rocprof.err; echo "stats rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/single -- python3 $R/tools/probes/probe_run.py 1 6 > $O/single.log 2>&1; echo "single rc=$?"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/single -name "*kernel_stats.csv" | head -1) $O/single_rig_kernel_stats.csv
if [ "${3:-}" != "nopmc" ]; then
  export PTZ_BA_STREAMS=1
  N=1000
  timeout 400 python3 $R/tools/probes/probe_c4pmc.py $N | tail -1   # (scene cache)
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc$i -- python3 $R/tools/probes/probe_c4pmc.py $N > $O/pmc$i.log 2>&1; echo "pmc$i [$grp] rc=$? $(tail -1 $O/pmc$i.log | cut -c1-80)"
  done
  cd $R; python3 profiles/summarize_pmc.py $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 $O/pmc6 > $O/pmc_summary.json
  LM=$(grep -o "lm_steps [0-9]*" $O/pmc1.log | tail -1 | cut -d' ' -f2)
  python3 profiles/make_traffic.py $O/pmc_summary.json --lm-steps ${LM:-17448} --tag C4:1000x200x500 > $O/traffic.json
  find $O -name "*counter_collection.csv" -delete
fi
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 tools/kstats.py $O/kernel_stats.csv | head -14; tail -1 $O/single.log; python3 tools/kstats.py $O/single_rig_kernel_stats.csv | head -14
