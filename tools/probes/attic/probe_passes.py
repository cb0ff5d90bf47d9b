"""Probe (not a test): per-pass wall time of the lock-step batch solve against the number of scenes still active."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scenes = [pkg.synth.make_scene(s, 200, 500) for s in range(16)]
scenes = [scenes[i % 16] for i in range(B)]
b = pkg.api.BaBatch(scenes); b.set_state(); b.solve()
os.environ["PTZ_BA_DEBUG_TIMING"] = "1"
b.solve()
