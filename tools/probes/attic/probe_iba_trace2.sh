#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u; cd $GRAFT_REPO_ROOT
# Probe (not a test): per-call timings of the cohort PTZ-IBA (PTZ_BATCHER_TRACE), summed
out=gpurun_out/${1:-iba_trace2}; mkdir -p $out
for g in 0 1; do
PTZ_BA_GPU_STRUCT=$g PTZ_BATCHER_TRACE=1 timeout 900 python tools/probes/probe_iba_batch.py 64 200 > $out/iba$g.txt 2> $out/trace$g.txt
tail -1 $out/iba$g.txt | cut -c1-200
n=$(grep -c "batcher ba" $out/trace$g.txt); half=$((n/3))
grep "batcher ba" $out/trace$g.txt | tail -$half | awk '{c+=$7; s+=$9; v+=$11; g+=$13; d+=$15; n++; p+=$3} END {print "GPU_STRUCT='$g' ba last run: calls",n,"create",c,"set",s,"solve",v,"get",g,"destroy",d}'
grep "batcher krt" $out/trace$g.txt | tail -1916 | awk '{p+=$7; c+=$9; d+=$11; n++} END {print "krt last run: n",n,"pack",p,"call",c,"device",d}'
done
