#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u; cd $GRAFT_REPO_ROOT
# Probe (not a test): per-call timings of the lock-step PTZ-IBA batch (PTZ_BATCHER_TRACE, PTZ_POOL_TRACE, PTZ_BA_DEBUG_TIMING).
out=gpurun_out/${1:-iba_trace}; mkdir -p $out
PTZ_BATCHER_TRACE=1 PTZ_POOL_TRACE=1 timeout 900 python tools/probes/probe_iba_batch.py 64 200 > $out/iba.txt 2> $out/trace.txt
tail -3 $out/iba.txt
grep "batcher krt" $out/trace.txt | tail -133 | awk '{p+=$7; c+=$9; d+=$11; n++} END {print "krt last run: n",n,"pack",p,"call",c,"device",d}'
grep "batcher ba" $out/trace.txt | tail -300 | awk '{c+=$7; s+=$9; v+=$11; g+=$13; d+=$15; n++} END {print "ba last calls: n",n,"create",c,"set",s,"solve",v,"get",g,"destroy",d}'
grep "batcher ba" $out/trace.txt | tail -12
echo "pool misses: $(grep -c 'ptzpool miss' $out/trace.txt)"; grep "ptzpool miss" $out/trace.txt | awk '{t[$3]+=$6; n[$3]++} END {for (k in t) print k, n[k], t[k], "ms"}'
grep "ptzpool miss" $out/trace.txt | tail -5
PTZ_BA_DEBUG_TIMING=1 PTZ_BATCHER_TRACE=1 timeout 900 python tools/probes/probe_iba_batch.py 16 200 2>&1 | grep "ptz_ba_create\|batcher ba" | tail -16
