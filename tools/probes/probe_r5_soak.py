"""Soak of the one-launch factorisation after the round-5 changes of its hand-overs: the C2 rig (13 tiles, 8 levels) solved N times,
then two rigs side by side; every solve must have the bits of the first (a hand-over gone wrong shows as different bits or PTZ_ENODEVICE)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
import __graft_entry__ as ge
pkg = ge.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for n in (1, 2):
    rigs = [pkg.synth.make_scene(3 + i, 200, 500) for i in range(n)]
    b = pkg.api.BaBatch(rigs); b.set_state()
    first, t0, launches = None, time.time(), 0
    for it in range(N):
        summ = b.solve(); cams, rays = b.get_state()
        launches += sum(s["num_lm_steps"] for s in summ)
        got = (summ, [c.copy() for c in cams], [r.copy() for r in rays])
        if first is None:
            first = got
            continue
        assert got[0] == first[0], (n, it)
        assert all(np.array_equal(a, c) for a, c in zip(got[1], first[1])) and all(np.array_equal(a, c) for a, c in zip(got[2], first[2])), (n, it)
    b.close()
    print(f"{n} rig(s): {N} solves, {launches} LM iterations, all with the bits of the first ({time.time() - t0:.1f} s)")
