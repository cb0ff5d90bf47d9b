"""Soak of the chain kernel's hand-overs after round 6 took their fences out (write-through stores + sc1 loads): every solve must have
the bits of the first.  Three settings: the C2 rig alone and two side by side (idle chip); the C2 rig while ANOTHER host thread keeps
the chip busy with relocalization launches and a 48-rig batch on streams of their own (uneven load: the case the guide says hides
stale hand-overs on an idle chip); twelve mid-size rigs in one launch (the many-systems form of the one-launch factorisation)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
import __graft_entry__ as ge
pkg = ge.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def soak(batch, n, tag):
    first, t0, its = None, time.time(), 0
    for it in range(n):
        summ = batch.solve(); cams, rays = batch.get_state()
        its += sum(s["num_lm_steps"] for s in summ)
        got = (summ, [c.copy() for c in cams], [r.copy() for r in rays])
        if first is None:
            first = got
            continue
        assert got[0] == first[0], (tag, it)
        assert all(np.array_equal(a, c) for a, c in zip(got[1], first[1])) and all(np.array_equal(a, c) for a, c in zip(got[2], first[2])), (tag, it)
    print(f"{tag}: {n} solves, {its} LM iterations, all with the bits of the first ({time.time() - t0:.1f} s)", flush=True)


for n in (1, 2):
    b = pkg.api.BaBatch([pkg.synth.make_scene(3 + i, 200, 500) for i in range(n)]); b.set_state()
    soak(b, N, f"{n} rig(s), idle chip")
    b.close()

# under load from another host thread
stop = threading.Event()
def load():
    rb = pkg.synth.make_reloc_queries(20000, 128, seed_id=5, factor_type=1)
    big = pkg.api.BaBatch([pkg.synth.make_scene(40 + i, 60, 200) for i in range(48)]); big.set_state()
    k = 0
    while not stop.is_set():
        pkg.api.krt_solve_batch(rb)
        if k % 3 == 0:
            big.solve()
        k += 1
    big.close()
th = threading.Thread(target=load); th.start()
time.sleep(1.0)
try:
    b = pkg.api.BaBatch([pkg.synth.make_scene(3, 200, 500)]); b.set_state()
    soak(b, N, "1 rig, chip loaded by another thread")
    b.close()
    b = pkg.api.BaBatch([pkg.synth.make_scene(3 + i, 200, 500) for i in range(2)]); b.set_state()
    soak(b, N // 2, "2 rigs, chip loaded by another thread")
    b.close()
finally:
    stop.set(); th.join()

b = pkg.api.BaBatch([pkg.synth.make_scene(60 + i, 70 + 5 * (i % 4), 250) for i in range(12)]); b.set_state()
soak(b, N, "12 rigs of 70-85 views in one launch")
b.close()
print("soak ok")
