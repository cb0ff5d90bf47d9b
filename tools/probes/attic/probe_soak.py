"""Probe (not a test): create / solve / destroy cycles over all factor types and shapes; device memory must stay flat."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
scenes = {ft: pkg.synth.make_scene(3 + ft, 120, 200, factor_type=ft) for ft in range(4)}
big = pkg.synth.make_scene(9, 200, 500)
free0 = None
t0 = time.perf_counter()
for it in range(240):
    ft = it % 4
    if ft == 3:
        pkg.api.ba_solve_disp(scenes[3])
    else:
        b = pkg.api.BaBatch([scenes[ft]] * (1 + it % 5)); b.set_state(); s = b.solve(); b.close()
        assert all(x["termination_type"] == 0 for x in s)
    if it % 20 == 0:
        pkg.api.ba_solve(big)
    if it == 40:
        free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print(f"240 cycles in {time.perf_counter() - t0:.1f} s; free device memory after cycle 40: {free0 / 2**20:.0f} MiB, at the end: {free1 / 2**20:.0f} MiB, drift {(free0 - free1) / 2**20:.1f} MiB")
pkg.api.lib().ptz_trim_cache()
print(f"after ptz_trim_cache: {torch.cuda.mem_get_info()[0] / 2**20:.0f} MiB free")
