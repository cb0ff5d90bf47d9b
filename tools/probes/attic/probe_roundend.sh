#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# what the driver runs at round end: GPU suite, smoke(), default bench
R=$GRAFT_REPO_ROOT; T=${1:-roundend}
mkdir -p $R/gpurun_out/$T; cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 2>&1 | tail -25 | tee gpurun_out/$T/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/$T/smoke.txt
timeout 1800 python bench.py ${BENCH_ARGS:-} > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/$T/bench.err
python3 - <<PY
import json
d=json.load(open('gpurun_out/$T/bench.json'))
print(json.dumps(d['headline'], indent=0))
print({k: d[k] for k in ('value','ms_per_step','n_gpus','vs_baseline')}); print(json.dumps(d['roofline'])); print(json.dumps(d.get('default_pipeline')))
c=d['c2_single_rig']; print(c['lm_iterations_per_s'], c['us_per_lm_iteration'], c['per_pass_critical_path_us'])
print(json.dumps(d.get('ptz_iba_batch'))); print(json.dumps(d.get('ptz_iba'))[:600]); print(json.dumps(d['c5_reloc'].get('cpu_baseline'))[:700])
print(json.dumps(d.get('cpu_baseline'))[:500])
for k,v in d['kernel_families'].items(): print(k, v.get('ms_per_solve'), v.get('frac'), v.get('achieved_TFLOPs'), v.get('traffic_ratio'))
PY
