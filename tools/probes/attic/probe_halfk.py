"""Probe (not a test): PTZ_BA_CHOL_HALFK=1 (column update with half operand tiles) reproduces the default's bits; child processes."""
import os, subprocess, sys, json
code = r'''
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
scenes = [pkg.synth.make_scene(20 + s, 40 + 12 * (s % 5), 150) for s in range(12)]
b = pkg.api.BaBatch(scenes); b.set_state(); summ = b.solve(); cams, rays = b.get_state(); b.close()
h = hashlib.sha256()
for c, r in zip(cams, rays): h.update(c.tobytes()); h.update(r.tobytes())
print(h.hexdigest(), sum(s["num_lm_steps"] for s in summ))
'''
out = []
for v in ("0", "1"):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PTZ_BA_CHOL_HALFK=v), capture_output=True, text=True)
    out.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
print(out, "IDENTICAL" if out[0] == out[1] else "DIFFERENT")
