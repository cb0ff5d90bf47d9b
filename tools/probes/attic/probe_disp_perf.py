"""Probe (not a test): PTZRayDistDisp on a C2-sized rig and in a batch of 64 -- time per LM iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
for n in (1, 64):
    scenes = [pkg.synth.make_scene(i, 200, 500, factor_type=3) for i in range(min(n, 4))]
    scenes = [scenes[i % len(scenes)] for i in range(n)]
    b = pkg.api.BaBatch(scenes); b.set_state(); s = b.solve()
    t = time.perf_counter(); s = b.solve(); dt = time.perf_counter() - t
    its = sum(x["num_lm_steps"] for x in s)
    print(f"DistDisp B={n}: {1e3*dt:.1f} ms per solve, {its} LM steps, {its/dt:.0f} it/s, terminations {sorted(set(x['termination_type'] for x in s))}")
    b.close()
sc = pkg.synth.make_scene(0, 200, 500, factor_type=3)
b = pkg.api.BaBatch([sc]); b.set_state(); b.solve()
b.set_profiling(True); s = b.solve(); prof = b.get_profile(); b.set_profiling(False)
print({k: (round(v["ms"], 2), v["launches"]) for k, v in prof.items() if v["launches"]}, "device ms", round(b.last_solve_ms(), 2), "steps", s[0]["num_lm_steps"])
