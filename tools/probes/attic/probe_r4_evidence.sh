#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# Round-4 evidence for profiles/: GPU suite + smoke + default bench (what the driver runs), then the kernel trace of the
# headline leg, the kernel trace of graph-replayed one-rig solves, and the stamps of the one-launch factorisation.
# usage: tools/probes/probe_r4_evidence.sh <tag> [notests]
R=$GRAFT_REPO_ROOT; T=${1:-r4_evidence}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
if [ "${2:-}" != "notests" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 2>&1 | tail -8 | tee $O/pytest.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
fi
timeout 1800 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 600 $O/bench.err
cd /tmp && export TMPDIR=/tmp
PTZ_BA_STREAMS=1 timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --headline-only --steps 2 > $O/bench_under_rocprof.json 2>$O/bench_under_rocprof.err; echo "stats rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/single -- python3 $R/tools/probes/probe_run.py 1 6 > $O/single.log 2>&1; echo "single rc=$?"
cd $R
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/single -name "*kernel_stats.csv" | head -1) $O/single_rig_kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +10M -delete
bash tools/probes/probe_r4_chain_stamps.sh chain_stamps > $O/chain_stamps.txt 2>&1; cat $O/chain_stamps.txt
python3 tools/kstats.py $O/kernel_stats.csv | head -14; tail -1 $O/single.log; python3 tools/kstats.py $O/single_rig_kernel_stats.csv | head -14
python3 - <<PY
import json
d=json.load(open('$O/bench.json'))
print({k: d[k] for k in ('value','ms_per_step','n_gpus','vs_baseline')}); print(json.dumps(d['roofline'])); print(json.dumps(d.get('default_pipeline')))
c=d['c2_single_rig']; print(c['lm_iterations_per_s'], c['us_per_lm_iteration'], c['per_pass_critical_path_us'])
print(json.dumps(d.get('ptz_iba_batch'))[:700]); print(json.dumps(d.get('ptz_iba'))[:500]); print(json.dumps(d.get('parity'))[:600])
for k,v in d['kernel_families'].items(): print(k, v.get('ms_per_solve'), v.get('frac'), v.get('achieved_TFLOPs'), v.get('traffic_ratio'))
PY
