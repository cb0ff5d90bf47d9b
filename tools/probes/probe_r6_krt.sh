#!/bin/bash
: "${GRAFT_REPO_ROOT:?run through gpurun}"; set -u
# round 6: the single-view LM on C5 (100 000 queries x 128 matches), both factor types, prev library against the working tree's; then the full-size parity test
R=$GRAFT_REPO_ROOT; cd $R
for l in prev product prev product; do
  if [ $l = product ]; then unset PTZCALIB_LIB; else export PTZCALIB_LIB=$R/tools/probes/hip/lib_$l.so; fi
  echo "== $l"; timeout 300 python - <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as ge
pkg = ge.load_package()
import numpy as np
for ft in (0, 1):
    rb = pkg.synth.make_reloc_queries(100000, 128, seed_id=1, factor_type=ft)
    best = 1e9
    for _ in range(4):
        _, summ, acc, ms = pkg.api.krt_solve_batch(rb)
        best = min(best, ms)
    print("factor", ft, "kernel ms", round(best, 3), "M queries/s", round(100000 / best / 1e3, 1), "accepted", int(acc.sum()), "iterations", sum(s["num_lm_steps"] for s in summ))
PY
done
unset PTZCALIB_LIB
[ "${TESTS:-1}" = "1" ] && timeout 1500 python -m pytest tests -m gpu -x -q -k "krt or c5 or reloc or incremental" 2>&1 | tail -4
