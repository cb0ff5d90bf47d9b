#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}"; set -u
# Round-3 evidence for profiles/: (1) C4 with this round's k_schur against round 2's kernels (PTZ_BA_SCHUR_W=1) on ONE box,
# per-family device times; (2) k_krt: kernel trace + counters for both lane forms, F and FDist, 100 000 queries.
R=$GRAFT_REPO_ROOT; T=${1:-r3_evidence}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
python3 - <<'PY' > $O/c4_ab.json
import json, os, sys, subprocess
out = {}
for w in ("0", "1"):
    env = dict(os.environ, PTZ_BA_SCHUR_W=w, PTZ_BA_STREAMS="1")
    r = subprocess.run([sys.executable, "tools/probes/probe_c4_families.py", "1000"], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    out["k_schur (round 3)" if w == "0" else "k_schur_w + W rows (round 2 kernels)"] = json.loads(line[-1]) if line else r.stderr[-400:]
print(json.dumps(out, indent=1))
PY
cat $O/c4_ab.json
cd /tmp && export TMPDIR=/tmp
for g in 16 64; do for ft in 0 1; do
  PTZ_KRT_GROUP=$g timeout 300 python3 $R/tools/probes/probe_krt.py 100000 $ft 2>/dev/null | head -1 | sed "s/^/lanes $g: /" | tee -a $O/krt_rates.txt
done; done
export PTZ_KRT_GROUP=16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/krt_stats -- python3 $R/tools/probes/probe_krt.py 100000 0 > /dev/null 2>&1; echo "krt stats rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/krt_pmc1 -- python3 $R/tools/probes/probe_krt.py 100000 0 > /dev/null 2>&1; echo "krt pmc1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/krt_pmc2 -- python3 $R/tools/probes/probe_krt.py 100000 0 > /dev/null 2>&1; echo "krt pmc2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/krt_pmc3 -- python3 $R/tools/probes/probe_krt.py 100000 0 > /dev/null 2>&1; echo "krt pmc3 rc=$?"
export PTZ_KRT_GROUP=64
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/krt64_pmc1 -- python3 $R/tools/probes/probe_krt.py 100000 0 > /dev/null 2>&1; echo "krt64 pmc1 rc=$?"
cd $R
python3 profiles/summarize_pmc.py $O/krt_pmc1 $O/krt_pmc2 $O/krt_pmc3 > $O/krt16_pmc_summary.json
python3 profiles/summarize_pmc.py $O/krt64_pmc1 > $O/krt64_pmc_summary.json
cp $(find $O/krt_stats -name "*kernel_stats.csv" | head -1) $O/krt_kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +10M -delete; find $O -name "*counter_collection.csv" -size +10M -delete
grep k_krt $O/krt_kernel_stats.csv | cut -c1-200; python3 -c "
import json
for f in ('krt16','krt64'):
    d=json.load(open('$O/%s_pmc_summary.json'%f)); print(f, json.dumps(d.get('k_krt')))"
