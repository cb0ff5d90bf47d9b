#!/bin/bash
# default bench line (as the driver runs it) + its copy for profiles/
R=$GRAFT_REPO_ROOT; T=${1:-final}
mkdir -p $R/gpurun_out/$T; cd $R
timeout 1500 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc=$?"
tail -c 400 gpurun_out/$T/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/$T/bench.json'))
print({k: d[k] for k in ('value','ms_per_step','n_gpus')}); print(d['config']['workload']); print(json.dumps(d['roofline'])); print(json.dumps(d.get('default_pipeline')))
print(json.dumps(d.get('c2_single_rig'), indent=0)[:2500]); print(json.dumps(d.get('cpu_baseline'))[:800])
for k,v in d['kernel_families'].items(): print(k, v)
"
