"""Probe (not a test): one profiled C4 solve (1000 distinct seeds) -> per-family device time, as one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
scenes = pkg.synth.make_scenes(range(n), 200, 500, cache_dir="/tmp/ptz_scene_cache")
b = pkg.api.BaBatch(scenes); b.set_state(); b.solve()
b.set_profiling(True); b.set_state(); summ = b.solve(); p = b.get_profile()
steps = sum(s["num_lm_steps"] for s in summ)
ms = b.last_solve_ms()
print(json.dumps(dict(scenes=n, lm_steps=steps, device_ms=round(ms, 2), lm_it_per_s=round(steps / ms * 1e3, 1),
                      family_ms={k: round(v["ms"], 2) for k, v in p.items()}, family_launches={k: v["launches"] for k, v in p.items()})))
