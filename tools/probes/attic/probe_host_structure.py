"""Probe (not a test; runs without a GPU): host-side structure stage of ptz_ba_batch_create over thread counts."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
views = int(sys.argv[1]) if len(sys.argv) > 1 else 170
scenes = pkg.synth.make_scenes(range(20), views, 500, cache_dir="/tmp/ptz_scene_cache")
lib = pkg.api.lib()
for n in (1, 5, 19):
    keep = []
    probs = (pkg.api.BaProblem * n)(*[pkg.api._pack_problem(s, keep) for s in scenes[:n]])
    for nt in (1, 2, 4, 8):
        ms = C.c_double()
        rc = lib.ptz_debug_host_structure(n, probs, nt, 5, C.byref(ms), None)
        assert rc == 0, rc
        print(f"{n} problems x {views} views, {nt} threads: {ms.value:.2f} ms per create-structure", flush=True)
