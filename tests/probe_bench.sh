#!/bin/bash
# new config tests + default bench line + group-count comparison at C4 size
R=$GRAFT_REPO_ROOT; T=${1:-bench}
mkdir -p $R/gpurun_out/$T
cd $R
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q 2>&1 | tail -15 > gpurun_out/$T/pytest_configs.txt
cat gpurun_out/$T/pytest_configs.txt
timeout 900 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/$T/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/$T/bench.json'))
print({k: d[k] for k in ('value','ms_per_step','n_gpus')}); print(d['config']['workload']); print(json.dumps(d['roofline'])); print(json.dumps(d.get('default_pipeline')))
print(json.dumps(d.get('c2_single_rig'), indent=0)[:1500]); print(json.dumps(d.get('cpu_baseline'))[:800])
for k,v in d['kernel_families'].items(): print(k, v)
"
for g in 1 2 3; do PTZ_BA_STREAMS=$g timeout 600 python bench.py --headline-only --scenes 1000 --distinct 32 --steps 2 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('profiled 1 group:', round(d['value']), 'ms', round(d['ms_per_step'],1))"; done
for g in 1 2 3; do PTZ_BA_STREAMS=$g timeout 600 python tests/probe_run.py 1000 2 2>&1 | tail -1; done
