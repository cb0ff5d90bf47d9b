#!/bin/bash
# Round artifacts: bench line, rocprofv3 kernel stats of the same command, PMC passes (traffic) on the same workload.
# usage: tests/probe_final.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r01}
mkdir -p $R/gpurun_out/$T
cd $R && timeout 600 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
cd /tmp && export TMPDIR=/tmp
# one scene group for everything that is profiled: every launch then covers the whole batch, as in the timed (profiled) region
# of bench.py, so that per-kernel averages and per-dispatch counters refer to the same launch shape
export PTZ_BA_STREAMS=1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats -- python3 $R/bench.py --headline-only > $R/gpurun_out/$T/bench_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/pmc1 -- python3 $R/tests/probe_run.py 256 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/$T/pmc2 -- python3 $R/tests/probe_run.py 256 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/$T/pmc3 -- python3 $R/tests/probe_run.py 256 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/$T/pmc4 -- python3 $R/tests/probe_run.py 256 1 > /dev/null 2>&1
# drop the big per-dispatch traces, keep summaries
find $R/gpurun_out/$T -name "*kernel_trace.csv" -size +20M -delete
ls -R $R/gpurun_out/$T | head -40; tail -c 400 $R/gpurun_out/$T/bench.json
