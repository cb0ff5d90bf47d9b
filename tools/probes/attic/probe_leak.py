"""Probe (not a test): repeated create / solve / destroy cycles and single-view launches; device memory after ptz_trim_cache()
must return to where it started (the pool parks blocks in between)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
free0, total = torch.cuda.mem_get_info()
scenes = [pkg.synth.make_scene(s, 12 + 4 * (s % 6), 80) for s in range(12)]
rb = pkg.synth.make_reloc_batch(64, 96, seed_id=1)
t = time.time()
for it in range(300):
    b = pkg.api.BaBatch(scenes[it % 7: it % 7 + 5]); b.set_state(); b.solve(); b.get_state(); b.close()
    pkg.api.ba_solve(scenes[it % 12])
    pkg.api.krt_solve_batch(rb)
    if it % 100 == 99:
        f, _ = torch.cuda.mem_get_info()
        print(f"iteration {it + 1}: {time.time() - t:.1f} s, device memory in use by the process tree: {(free0 - f) / 2**20:.1f} MiB", flush=True)
pkg.api.trim_cache()
f, _ = torch.cuda.mem_get_info()
print(f"after trim_cache: {(free0 - f) / 2**20:.1f} MiB above the start")
