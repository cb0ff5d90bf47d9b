"""Probe (not a test): ptz_ba_batch_create for IBA-sized batches, alone and from several host threads at once."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
scenes = pkg.synth.make_scenes(range(20), 170, 500, cache_dir="/tmp/ptz_scene_cache")

def cycle(sc, out, reps=6):
    ts = []
    for rep in range(reps):
        t0 = time.perf_counter(); b = pkg.api.BaBatch(sc); t1 = time.perf_counter()
        b.set_state(); t2 = time.perf_counter(); b.solve(); t3 = time.perf_counter(); b.get_state(); t4 = time.perf_counter(); b.close(); t5 = time.perf_counter()
        ts.append([1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4)])
    out.append(np.array(ts)[2:].mean(0))

for n in (1, 5, 19):
    out = []; cycle(scenes[:n], out)
    print(f"one thread, {n} problems: create {out[0][0]:.2f} set {out[0][1]:.2f} solve {out[0][2]:.2f} get {out[0][3]:.2f} destroy {out[0][4]:.2f} ms", flush=True)
for nth in (2, 4):
    out = []
    th = [threading.Thread(target=cycle, args=(scenes[5 * k:5 * k + 5], out)) for k in range(nth)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    m = np.array(out).mean(0)
    print(f"{nth} threads x 5 problems: create {m[0]:.2f} set {m[1]:.2f} solve {m[2]:.2f} get {m[3]:.2f} destroy {m[4]:.2f} ms; wall per cycle {(time.perf_counter()-t0)/6*1e3:.2f} ms", flush=True)
os.environ["PTZ_BA_DEBUG_TIMING"] = "1"
for n in (5, 19):
    b = pkg.api.BaBatch(scenes[:n]); b.close()
