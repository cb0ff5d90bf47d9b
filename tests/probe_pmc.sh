#!/bin/bash
# usage: tests/probe_pmc.sh <tag> <B>   -> gpurun_out/pmc_<tag>/{p1,p2,p3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; B=${2:-64}
mkdir -p $R/gpurun_out/pmc_$T
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_$T/p1 -- python3 $R/tests/probe_run.py $B 1 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_$T/p2 -- python3 $R/tests/probe_run.py $B 1 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_$T/p3 -- python3 $R/tests/probe_run.py $B 1 > /dev/null 2>&1
find $R/gpurun_out/pmc_$T -name "*counter_collection.csv" | head
