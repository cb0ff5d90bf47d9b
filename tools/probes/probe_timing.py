"""Timing probe (not a test): per-kernel-family device time of the BA solve at several batch sizes."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
import numpy as np

def run(B, nv=200, opv=500, reps=2, prof=True):
    t = time.time()
    scenes = [pkg.synth.make_scene(s, nv, opv) for s in range(min(B, 4))]
    scenes = [scenes[i % len(scenes)] for i in range(B)]
    tg = time.time() - t
    t = time.time()
    b = pkg.api.BaBatch(scenes); b.set_state()
    tc = time.time() - t
    b.solve()
    out = []
    for r in range(reps):
        t = time.time(); summ = b.solve(); wall = time.time() - t
        its = sum(s["num_lm_steps"] for s in summ)
        out.append((wall, b.last_solve_ms(), its))
    res = dict(B=B, gen_s=tg, create_s=tc, wall_s=[o[0] for o in out], dev_ms=[o[1] for o in out], lm_steps=out[0][2],
               it_per_s=out[-1][2] / out[-1][0], term=[s["termination_type"] for s in summ][:4])
    if prof:
        b.set_profiling(True); b.solve(); p = b.get_profile(); b.set_profiling(False)
        res["profile_ms"] = {k: round(v["ms"], 3) for k, v in p.items() if v["launches"]}
        res["profile_n"] = {k: v["launches"] for k, v in p.items() if v["launches"]}
    b.close()
    return res

for B in [int(x) for x in sys.argv[1:]] or [1, 8]:
    print(json.dumps(run(B)), flush=True)
