"""Probe (not a test): the one-launch factorisation for batches of 3-6 SMALL systems (rigs of 60-110 views, as the incremental
pipeline's rounds solve them): device ms per solve with it (default rule: all tiles on the chip at once) and with PTZ_BA_CHOL_CHAIN_MAX=2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
for views, B in ((60, 6), (80, 4), (110, 3), (110, 4)):
    scenes = [pkg.synth.make_scene(70 + s, views, 300) for s in range(B)]
    b = pkg.api.BaBatch(scenes); b.set_state(); b.solve()
    ms = []
    for r in range(3):
        s = b.solve(); ms.append(b.last_solve_ms())
    its = sum(x["num_lm_steps"] for x in s)
    print(f"{B} rigs x {views} views: {min(ms):.3f} ms per solve, {its} LM steps", flush=True)
    b.close()
