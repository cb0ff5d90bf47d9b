"""Probe (not a test): the one-launch factorisation under concurrency -- host threads solving one- and two-rig batches side by
side for a while (several chol_chain_kernel launches share the chip: tickets, generations and bounded waits at work), every
result compared bit for bit with the serial solve of the same scene."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
N_THREADS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
MAXG = int(sys.argv[3]) if len(sys.argv) > 3 else 2   # largest batch (more than 8: the mid-size form of the one-launch factorisation)
scenes = [pkg.synth.make_scene(40 + s, 60 + 20 * (s % 4), 200 + 50 * (s % 3)) for s in range(6)]
ref = [pkg.api.ba_solve(sc) for sc in scenes]
bad = []
def work(tid):
    rng = np.random.default_rng(tid)
    for r in range(ROUNDS):
        i = int(rng.integers(len(scenes))); j = int(rng.integers(len(scenes)))
        ids = [i] if (r + tid) % 2 else [i, j]
        if MAXG > 2 and r % 3 == 0: ids = [int(x) for x in rng.integers(len(scenes), size=int(rng.integers(9, MAXG + 1)))]
        group = [scenes[x] for x in ids]
        b = pkg.api.BaBatch(group); b.set_state()
        for rep in range(2):  # the second solve replays the recorded graph
            summ = b.solve(); cams, rays = b.get_state()
            for k, idx in enumerate(ids):
                if not (np.array_equal(cams[k], ref[idx][0]) and np.array_equal(rays[k], ref[idx][1]) and summ[k] == ref[idx][2]):
                    bad.append((tid, r, rep, idx, summ[k]["termination_type"]))
        b.close()
t0 = time.perf_counter()
ths = [threading.Thread(target=work, args=(t,)) for t in range(N_THREADS)]
[t.start() for t in ths]; [t.join() for t in ths]
n = N_THREADS * ROUNDS * 2
print(f"{N_THREADS} threads x {ROUNDS} rounds: {n} solves of batches of 1 to {MAXG} rigs in {time.perf_counter() - t0:.1f} s, {len(bad)} results differ from the serial solve {bad[:5]}")
sys.exit(1 if bad else 0)
