"""BASELINE.json configurations at FULL size on one MI355X, through the C-ABI (-m gpu).

  C3  configs[2]: WorldCup14-shaped batch, four matches of different size     (reference loop: run_ptzba_worldcup14.sh:4-7; the
                  data set is not available, so four synthetic broadcast-camera scenes stand in: 1280 x 720, +-60 degrees of pan)
  C4  configs[3]: synthetic 1000-scene batch, 200 views x 500 obs/view each  (reference loop: scripts/run_ptzba_synthetic.sh:4-13)
  C5  configs[4]: 100 000 relocalization queries x 128 matches                (reference loop: src/app/run_ptz_reloc.cc:68-118)

The oracle cannot solve these in seconds, so parity is checked on samples (same seeded inputs, the oracle in its
reference-faithful numeric-differentiation mode, tolerance 1e-6 relative as BASELINE.json's north_star states) and through
size-independent properties: every scene of the batch has the bits of its solo solve (LM control flow sits on thresholds, so
this is the strongest statement that the batch machinery changes nothing), scenes built from the same seed inside the batch
agree bit for bit, and every relocalization gate is evaluated for every query.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _relative_rotations(orc, cam):
    R = [orc.rodrigues(c[4:7]) for c in cam]
    return np.stack([r @ R[0].T for r in R])


def _rodrigues_many(rv):
    """cv::Rodrigues for an [n, 3] array of rotation vectors -> [n, 3, 3] (numpy restatement for the whole-population checks)."""
    th = np.linalg.norm(rv, axis=1)
    safe = np.where(th < 2.220446049250313e-16, 1.0, th)
    k = rv / safe[:, None]
    c, s_ = np.cos(th), np.sin(th)
    K = np.zeros((len(rv), 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    R = c[:, None, None] * np.eye(3)[None] + (1 - c)[:, None, None] * (k[:, :, None] * k[:, None, :]) + s_[:, None, None] * K
    R[th < 2.220446049250313e-16] = np.eye(3)
    return R


def c3_standin_scenes(pkg):
    """Four matches of different size, as run_ptzba_worldcup14.sh:4-7 runs four recordings one after the other: 1280 x 720
    (eval_worldcup.py:68-69), a broadcast camera's +-60 degrees of pan, the distortion model the tool uses there (--dist)."""
    shapes = [(60, 300), (48, 250), (72, 350), (55, 280)]
    return [pkg.synth.make_scene(100 + i, nv, opv, factor_type=1, width=1280, height=720, pan_range_deg=120.0)
            for i, (nv, opv) in enumerate(shapes)]


def test_c3_standin_batch(pkg, orc):
    """BASELINE configs[2] as a BATCH: the four heterogeneous scenes in ONE ptz_ba_batch and through ptz_ba_solve_sharded's
    longest-first dealing (the GPU of this box listed twice, then four times: what an 8-GPU node does with four matches).  Every
    scene has the bits of its solo solve on every path, and agrees with the reference-faithful numeric-differentiation oracle:
    same termination, iteration and accepted-step counts, cost to 1e-9, focal lengths, k1 and relative rotations to 1e-6."""
    scenes = c3_standin_scenes(pkg)
    assert len({(s.n_cam, s.n_obs) for s in scenes}) == 4
    solo = [pkg.api.ba_solve(sc) for sc in scenes]
    b = pkg.api.BaBatch(scenes); b.set_state(); summ = b.solve(); cams, rays = b.get_state(); b.close()
    for i in range(4):
        assert summ[i] == solo[i][2] and np.array_equal(cams[i], solo[i][0]) and np.array_equal(rays[i], solo[i][1])
    for devs in ([0, 0], [0, 0, 0, 0]):
        c2, r2, s2 = pkg.api.ba_solve_sharded(scenes, devs)
        for i in range(4):
            assert s2[i] == solo[i][2] and np.array_equal(c2[i], solo[i][0]) and np.array_equal(r2[i], solo[i][1])
    for i, sc in enumerate(scenes):
        ocam, _, _, osumm, _ = orc.ba_solve(sc, jacobian_mode=orc.JAC_NUMERIC, num_threads=orc.usable_cores())
        assert summ[i]["termination_type"] == osumm["termination_type"] == 0
        assert summ[i]["num_iterations"] == osumm["num_iterations"]
        assert summ[i]["num_successful_steps"] == osumm["num_successful_steps"]
        assert abs(summ[i]["final_cost"] - osumm["final_cost"]) / osumm["final_cost"] < 1e-9
        assert _rel(cams[i][:, 0], ocam[:, 0]) < 1e-6
        assert np.abs(cams[i][:, 10] - ocam[:, 10]).max() < 1e-6
        assert np.abs(_relative_rotations(orc, cams[i]) - _relative_rotations(orc, ocam)).max() < 1e-6


def test_c4_full_batch(pkg, orc):
    """bench.py's own workload under test: 1000 C2-shaped scenes, every one its own seed, in ONE batch on one GPU.  All
    converge; focal lengths come back at noise level for every scene; a sample of 8 scenes is bit-equal to its solo solve; 16
    scenes spread over the LM-step histogram (shortest to longest) agree with the numeric-diff oracle (termination, iteration
    and accepted-step counts, cost to 1e-9, focal lengths and gauge-invariant relative rotations within 1e-6).  (That copies of one scene inside a batch take identical trajectories is test_c4_cycled_seeds.)"""
    n = 1000
    scenes = pkg.synth.make_scenes(range(n), 200, 500)
    b = pkg.api.BaBatch(scenes)
    b.set_state()
    summ = b.solve()
    cams, rays = b.get_state()
    b.close()
    assert len(summ) == n
    assert all(s["termination_type"] == 0 for s in summ), "every scene of C4 converges"
    its = np.array([s["num_lm_steps"] for s in summ])
    assert its.min() >= 3 and its.max() < 200
    # Focal lengths come back at noise level -- except where the ALGORITHM ends in a poor local minimum: scene 609 of this seed
    # stream does (121 LM steps, the straggler of the batch), in the oracle exactly as on the device.  Such scenes are held to
    # the oracle instead of to the ground truth.
    ferr = np.array([np.abs(cams[i][:, 0] - scenes[i].cam_gt[:, 0]).mean() for i in range(n)])
    off = [int(i) for i in np.flatnonzero(ferr >= 2.5)]
    assert len(off) <= 3, off
    for k in off:
        ocam, _, _, osumm, _ = orc.ba_solve(scenes[k], jacobian_mode=orc.JAC_ANALYTIC, num_threads=orc.usable_cores())
        assert summ[k]["num_iterations"] == osumm["num_iterations"] and summ[k]["num_successful_steps"] == osumm["num_successful_steps"]
        assert abs(summ[k]["final_cost"] - osumm["final_cost"]) / osumm["final_cost"] < 1e-9
        assert _rel(cams[k][:, 0], ocam[:, 0]) < 1e-6
    sample = [0, 1, 7, 113, 429, 631, 847, 999]
    for k in sample:
        cam, ray, s = pkg.api.ba_solve(scenes[k])
        assert s == summ[k]
        assert np.array_equal(cam, cams[k]) and np.array_equal(ray, rays[k])
    # the reference-faithful (numeric-differentiation) oracle on 16 scenes spread over the batch's LM-step histogram, from the
    # shortest solve to the longest (the straggler): LM control flow sits on thresholds, so the long trajectories are where an
    # arithmetic difference would show first
    order = np.argsort(its, kind="stable")
    spread = sorted({int(order[i]) for i in np.linspace(0, n - 1, 16).round().astype(int)})
    assert int(order[-1]) in spread and int(order[0]) in spread and len(spread) >= 12
    worst = {"f": 0.0, "rot": 0.0, "cost": 0.0}
    sensitive = []
    for k in spread:
        ocam, _, _, osumm, _ = orc.ba_solve(scenes[k], jacobian_mode=orc.JAC_NUMERIC, num_threads=orc.usable_cores())
        if summ[k]["num_iterations"] != osumm["num_iterations"]:
            # The two modes of the ORACLE take different numbers of steps on this scene (the 1e-8 noise of central differences moves a
            # long trajectory through a flat valley: the batch's straggler, which ends in a poor local minimum).  That is a property
            # of the algorithm, not of the device: such a scene is held to the closed-form oracle -- whose trajectory the device must
            # reproduce step for step -- and listed.
            nsteps = osumm["num_iterations"]
            ocam, _, _, osumm, _ = orc.ba_solve(scenes[k], jacobian_mode=orc.JAC_ANALYTIC, num_threads=orc.usable_cores())
            sensitive.append((k, int(summ[k]["num_iterations"]), int(nsteps)))
        assert summ[k]["termination_type"] == osumm["termination_type"], k
        assert summ[k]["num_iterations"] == osumm["num_iterations"], (k, summ[k]["num_iterations"], osumm["num_iterations"])
        assert summ[k]["num_successful_steps"] == osumm["num_successful_steps"], k
        worst["cost"] = max(worst["cost"], abs(summ[k]["final_cost"] - osumm["final_cost"]) / osumm["final_cost"])
        worst["f"] = max(worst["f"], _rel(cams[k][:, 0], ocam[:, 0]))
        worst["rot"] = max(worst["rot"], float(np.abs(_relative_rotations(orc, cams[k]) - _relative_rotations(orc, ocam)).max()))
    print(f"C4 oracle sample {spread}: LM steps {[int(its[k]) for k in spread]}, worst {worst}; scenes on which the oracle's own two "
          f"Jacobian modes take different step counts (device, numeric oracle): {sensitive}")
    assert worst["cost"] < 1e-9 and worst["f"] < 1e-6 and worst["rot"] < 1e-6, worst
    # which scenes were held to the closed-form oracle instead of the reference-faithful one is PINNED: scene 609 (the straggler of this
    # seed stream, 121 LM steps through a flat valley) is the only one known to be sensitive.  Any other scene turning up here is a change
    # of the device's (or the oracle's) arithmetic and must be looked at, not absorbed.
    assert {k for k, _, _ in sensitive} <= {609} and all(its[k] > 60 for k, _, _ in sensitive), sensitive


def test_c4_cycled_seeds(pkg):
    """Copies of a scene inside one large batch (16 seeds dealt round-robin over 256 slots, two scene groups, compacted tail
    passes) take identical trajectories and have the bits of the solo solve."""
    n, distinct = 256, 16
    base = pkg.synth.make_scenes(range(distinct), 200, 500)
    b = pkg.api.BaBatch([base[i % distinct] for i in range(n)]); b.set_state(); summ = b.solve(); cams, rays = b.get_state(); b.close()
    for k in range(distinct):
        for j in range(k + distinct, n, distinct):
            assert summ[j] == summ[k] and np.array_equal(cams[j], cams[k]) and np.array_equal(rays[j], rays[k])
    for k in (0, 5, 15):
        cam, ray, s = pkg.api.ba_solve(base[k])
        assert s == summ[k] and np.array_equal(cam, cams[k]) and np.array_equal(ray, rays[k])


@pytest.mark.parametrize("ftype", [0, 1])
def test_c5_full_reloc(pkg, orc, ftype):
    """100 000 queries x 128 matches in one launch (F and FDist, the two factor types the reference's tools use): every query
    gets a termination type and every gate of KRTOptimizer::CheckResults (krt_optimizer.cc:504-533) an answer; ALL
    100 000 agree with the oracle's numeric-diff + QR solve in termination type, iteration count and gate decision (accepted
    cameras to 1e-6); a sub-batch of the same queries reproduces the bits."""
    n = 100000
    rb = pkg.synth.make_reloc_queries(n, 128, seed_id=7 + ftype, factor_type=ftype)
    cam_w, summ, acc, ms = pkg.api.krt_solve_batch(rb)
    assert cam_w.shape == (n, 15) and len(summ) == n
    term = np.array([s["termination_type"] for s in summ])
    assert set(np.unique(term)) <= {0, 1, 2}
    assert set(np.unique(acc)) <= {0, 1}
    assert np.all(acc[term != 0] == 0)                 # only CONVERGENCE can pass the gates
    rej = acc == 0
    assert np.array_equal(cam_w[rej], rb.cam_init[rej])  # outputs untouched on failure (krt_optimizer.cc:396-403)
    assert acc.mean() > 0.95
    ok = acc == 1
    assert np.median(np.abs(cam_w[ok, 0] / rb.cam_gt[ok, 0] - 1)) < 2e-3   # focal recovered at noise level
    # the oracle (numeric differentiation over all 15 parameters + Householder QR, run_ptz_reloc.cc:68-118 as oracle/ptz_oracle.c
    # restates it) on ALL queries: termination type, iteration count and gate decision of every single one, and the accepted
    # cameras to the north-star tolerance
    ocam, osumm, oacc = orc.krt_solve_batch(rb, num_threads=orc.usable_cores(), jacobian_mode=orc.JAC_NUMERIC)
    oterm = np.array([s["termination_type"] for s in osumm])
    oits = np.array([s["num_iterations"] for s in osumm])
    its = np.array([s["num_iterations"] for s in summ])
    bad = np.flatnonzero((term != oterm) | (its != oits) | (acc != oacc))
    print(f"C5 ftype {ftype}: {n} queries against the oracle: {len(bad)} differ in termination / iterations / gate"
          + (f" (first: {[(int(q), int(term[q]), int(oterm[q]), int(its[q]), int(oits[q]), int(acc[q]), int(oacc[q])) for q in bad[:8]]})" if len(bad) else ""))
    assert len(bad) == 0
    both = (acc == 1)
    assert np.abs(cam_w[both, 0] / ocam[both, 0] - 1).max() < 1e-6
    assert np.abs(_rodrigues_many(cam_w[both, 4:7]) - _rodrigues_many(ocam[both, 4:7])).max() < 1e-6
    if ftype & 1:
        assert np.abs(cam_w[both, 10] - ocam[both, 10]).max() < 1e-6
    # the same queries as a smaller launch: identical bits (a query's result does not depend on its neighbours).  The lane
    # form is part of the bits (16 or 64 lanes sum a query's rows in a different order), and the automatic choice depends on
    # the launch size, so the small launch asks for the form the large one got; the other form agrees to rounding.
    import copy
    m = 5000
    sub = copy.copy(rb)
    sub.n_query = m
    sub.match_ptr = rb.match_ptr[:m + 1]
    sub.uv_ref = rb.uv_ref[:rb.match_ptr[m]]; sub.uv_cur = rb.uv_cur[:rb.match_ptr[m]]
    sub.cam_ref = rb.cam_ref[:m]; sub.cam_init = rb.cam_init[:m]; sub.cam_gt = rb.cam_gt[:m]
    cam2, summ2, acc2, _ = pkg.api.krt_solve_batch(sub, krt_lanes_per_query=16)
    assert np.array_equal(cam2, cam_w[:m]) and np.array_equal(acc2, acc[:m]) and summ2 == summ[:m]
    # the other lane form sums a query's rows in another order: reported, and held to the same oracle -- every query of the
    # sub-launch that takes another decision than the 16-lane form is listed, none may disagree with the oracle
    cam3, summ3, acc3, _ = pkg.api.krt_solve_batch(sub, krt_lanes_per_query=64)
    its3 = np.array([s["num_iterations"] for s in summ3]); term3 = np.array([s["termination_type"] for s in summ3])
    flips = np.flatnonzero((acc3 != acc[:m]) | (its3 != its[:m]) | (term3 != term[:m]))
    print(f"C5 ftype {ftype}: 64-lane form on the first {m} queries: {len(flips)} decisions differ from the 16-lane form {[int(q) for q in flips[:8]]}")
    assert np.array_equal(acc3, oacc[:m]) and np.array_equal(its3, oits[:m]) and np.array_equal(term3, oterm[:m])
    same = (acc3 == 1) & (acc[:m] == 1)
    assert np.abs(cam3[same, 0] / cam_w[:m][same, 0] - 1).max() < 1e-6


def test_nccl_backend_world_size_one(pkg, tmp_path):
    """The multi-GPU bookkeeping of bench.py on the real RCCL backend with ONE rank (all this pool's boxes have): launched
    through torch.distributed.run as the driver launches it; the line must carry the world size RCCL reports and a
    result gather that went through the collective."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import socket
    with socket.socket() as sk:  # a free port (a fixed one collides with a concurrent run or a socket in TIME_WAIT)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--scenes", "8",
           "--views", "20", "--obs", "100", "--headline-only"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["converged_scenes"] == 8
    assert d["parallel"]["backend"] == "nccl" and d["parallel"]["world_size_seen_by_collective"] == 1
    assert d["parallel"]["gather_ms"] is not None and d["parallel"]["ranks_ms_per_step"] is not None


@pytest.mark.parametrize("scaling,scenes,want_total,want_rank0", [("weak", 3, 6, 3), ("strong", 5, 5, 3)])
def test_two_ranks_on_the_real_kernels(pkg, tmp_path, scaling, scenes, want_total, want_rank0):
    """N > 1 on the real kernels without an 8-GPU node: bench.py under torch.distributed.run with TWO ranks that share device 0
    (PTZ_BENCH_SHARED_GPU=1: gloo carries the collectives; a debugging mode, not a measurement).  Checked: the launcher starts
    before anything touches the GPU (it is a fresh process tree), the collective sees two ranks, rank 1's scenes start where rank 0's
    end (weak: 3 + 3 scenes, strong: 5 scenes dealt 3 + 2), and every gathered 15 x n_cam block -- rank 1's included -- has the bits
    of a solo solve of that scene: the sharding moved whole problems and the gather put them back in scene order."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["PTZ_BENCH_SHARED_GPU"] = "1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dump = str(tmp_path / "gathered.npy")
    n_views, n_obs = 24, 100
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--scenes", str(scenes),
           "--views", str(n_views), "--obs", str(n_obs), "--scaling", scaling, "--headline-only", "--scene-cache", "", "--dump-gathered", dump]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    par = d["parallel"]
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["vs_baseline"] is None
    assert par["world_size"] == 2 and par["world_size_seen_by_collective"] == 2 and par["backend"] == "gloo"
    assert len(par["ranks_ms_per_step"]) == 2 and par["gather_ms"] is not None
    assert d["scenes_total"] == want_total and d["converged_scenes"] == want_total
    assert par["first_scene_of_rank0"] == 0 and par["scenes_of_rank0"] == want_rank0
    g = np.load(dump)
    assert g.shape == (want_total, 15 * n_views + 3)
    for sid in range(want_total):  # scene ids are the seeds: rank 1 owns [want_rank0, want_total)
        cam, _, summ = pkg.api.ba_solve(pkg.synth.make_scene(sid, n_views, n_obs))
        assert np.array_equal(g[sid, :15 * n_views].reshape(n_views, 15), cam), sid
        assert g[sid, 15 * n_views] == summ["termination_type"] and g[sid, 15 * n_views + 1] == summ["num_iterations"]
        assert g[sid, 15 * n_views + 2] == summ["final_cost"]


def _run_bench_ranks(n_ranks, extra, timeout=900):
    """bench.py under torch.distributed.run with n_ranks ranks sharing device 0 (PTZ_BENCH_SHARED_GPU=1, gloo): the parsed JSON line."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["PTZ_BENCH_SHARED_GPU"] = "1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(n_ranks), "--headline-only", "--scene-cache", ""] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_c5_as_headline_on_two_ranks(pkg):
    """`bench.py --config C5 --gpus 2` (the query-sharded branch, which no other test runs): every rank solves ITS queries
    (seed 1 + rank), the line counts the queries and LM iterations of both ranks over the slowest rank's time, and the figures of
    rank 0's launch are those of the same queries solved through the batch API."""
    nq, steps = 3000, 2
    d = _run_bench_ranks(2, ["--config", "C5", "--queries", str(nq), "--steps", str(steps), "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["metric"].startswith("LM iterations/sec (PTZ-Reloc") and d["config"]["queries_per_gpu"] == nq
    assert d["config"]["parallelism"] == "query-sharded x2"
    par = d["parallel"]
    assert par["world_size"] == 2 and par["world_size_seen_by_collective"] == 2 and len(par["ranks_ms_per_step"]) == 2
    t = 1e-3 * d["ms_per_step"] * steps
    assert abs(d["queries_per_s"] * t - 2 * nq * steps) < 1e-6 * nq  # both ranks' queries over the slowest rank's time
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["kernel"] == "k_krt" and 0 < d["roofline"]["frac"] < 1
    # LM iterations of BOTH ranks: rank r solves the queries of seed 1 + r
    its = 0
    for r in range(2):
        rb = pkg.synth.make_reloc_queries(nq, 128, seed_id=1 + r, factor_type=0)
        _, summ, _, _ = pkg.api.krt_solve_batch(rb)
        its += sum(x["num_lm_steps"] for x in summ)
    assert abs(d["value"] * t - its * steps) <= 1e-6 * its * steps


def test_bench_c2_as_headline_on_two_ranks(pkg, tmp_path):
    """`bench.py --config C2 --gpus 2`: ONE rig per rank (launch shapes of one system: the one-launch factorisation, the fused
    control), rank 1's rig is seed 1, and both gathered blocks have the bits of solo solves."""
    dump = str(tmp_path / "gathered.npy")
    n_views, n_obs = 40, 120
    d = _run_bench_ranks(2, ["--config", "C2", "--steps", "1", "--warmup", "0", "--views", str(n_views), "--obs", str(n_obs), "--dump-gathered", dump])
    assert d["n_gpus"] == 2 and d["scenes_total"] == 2 and d["converged_scenes"] == 2
    assert d["config"]["scenes_per_gpu"] == 1 and d["config"]["workload"].startswith("C2 (BASELINE configs[1])")
    g = np.load(dump)
    assert g.shape == (2, 15 * n_views + 3)
    its = 0
    for sid in range(2):
        cam, _, summ = pkg.api.ba_solve(pkg.synth.make_scene(sid, n_views, n_obs))
        assert np.array_equal(g[sid, :15 * n_views].reshape(n_views, 15), cam), sid
        assert g[sid, 15 * n_views + 1] == summ["num_iterations"] and g[sid, 15 * n_views + 2] == summ["final_cost"]
        its += summ["num_lm_steps"]
    assert abs(d["value"] * 1e-3 * d["ms_per_step"] - its) < 1e-6 * its


def test_eight_ranks_strong_scaling_uneven_shards(pkg, tmp_path):
    """The launch the driver's scaling run makes at N = 8, on one shared GPU: 1003 small scenes dealt to eight ranks (three shards of
    126, five of 125), gathered in scene order -- first contact with `--gpus 8` must not be the driver's."""
    dump = str(tmp_path / "gathered.npy")
    n_views, n_obs, total = 12, 60, 1003
    d = _run_bench_ranks(8, ["--scaling", "strong", "--scenes", str(total), "--steps", "1", "--warmup", "0", "--views", str(n_views), "--obs", str(n_obs),
                             "--workers", "2", "--dump-gathered", dump], timeout=1500)
    par = d["parallel"]
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["scenes_total"] == total
    assert par["world_size_seen_by_collective"] == 8 and len(par["ranks_ms_per_step"]) == 8
    assert par["first_scene_of_rank0"] == 0 and par["scenes_of_rank0"] == 126
    g = np.load(dump)
    assert g.shape == (total, 15 * n_views + 3)
    for sid in (0, 125, 126, 251, 252, 377, 378, 502, 503, 627, 628, 752, 753, 877, 878, 1002):  # both sides of every shard boundary
        cam, _, summ = pkg.api.ba_solve(pkg.synth.make_scene(sid, n_views, n_obs))
        assert np.array_equal(g[sid, :15 * n_views].reshape(n_views, 15), cam), sid
        assert g[sid, 15 * n_views] == summ["termination_type"] and g[sid, 15 * n_views + 1] == summ["num_iterations"]


def test_ptz_iba_batch_takes_the_decisions_of_solo_runs(pkg, monkeypatch):
    """PtzIncrementalOptimizer::SolveBatch (host/device_batcher.h): rigs of different size in lock step -- all pending bundle
    adjustments of a round in ONE ptz_ba_batch, all pending registration attempts in ONE ptz_krt_solve_batch launch -- against
    the same rigs calibrated one after the other (the reference's loop, run_ptzba_synthetic.sh:4-13): identical event
    sequences (seed pair, every registration and the reference it was made against, every bundle adjustment with its iteration
    count), identical registered sets, bit-identical cameras.  And the batching is real: fewer library calls than solo runs make."""
    shapes = [(1, 20, True), (3, 24, False), (6, 40, True), (2, 20, True), (9, 32, True)]
    tables, cam0 = [], []
    for seed, n_views, bi in shapes:
        sc = pkg.synth.make_scene(seed, n_views, 100)
        tb = pkg.synth.make_match_table(sc, bidirectional=bi)
        tables.append(tb)
        c = np.zeros((tb.n_img, 15)); c[:, 0] = c[:, 1] = 1.0
        cam0.append(c)
    solo = [pkg.hostlib.incremental_solve(tb, c, max_iter=200) for tb, c in zip(tables, cam0)]
    batch, stats = pkg.hostlib.incremental_solve_batch(tables, cam0, max_iter=200)
    for a, b in zip(solo, batch):
        assert a["ok"] and b["ok"]
        assert a["events"] == b["events"]
        assert a["registered"] == b["registered"] and a["lm_iterations"] == b["lm_iterations"]
        assert np.array_equal(a["cameras"], b["cameras"])
    n_ba_solo = sum(sum(1 for e in a["events"] if e[0] == 2) for a in solo)
    assert stats["ba_problems"] == n_ba_solo and stats["ba_batches"] < 0.5 * n_ba_solo
    assert stats["krt_launches"] <= stats["rounds"]
    # several cohorts (independent lock steps side by side, SolveBatch's default for many rigs) and no batching at all (one
    # cohort per rig: concurrent solo runs on one device): the same decisions and bits
    for cohorts in ("2", "3", "5"):
        monkeypatch.setenv("PTZ_IBA_COHORTS", cohorts)
        again, st2 = pkg.hostlib.incremental_solve_batch(tables, cam0, max_iter=200)
        for a, b in zip(solo, again):
            assert b["ok"] and a["events"] == b["events"] and a["registered"] == b["registered"] and np.array_equal(a["cameras"], b["cameras"])
        assert st2["ba_problems"] == n_ba_solo


def test_ptz_iba_over_resident_tracks_takes_the_decisions_of_host_packed_runs(pkg, monkeypatch):
    """PtzIncrementalOptimizer keeps a rig's tracks resident on the device and asks for every bundle adjustment as a VIEW of them
    (ptz_ba_batch_create_views builds the packed problem there, Pix2Ray included); PTZ_IBA_VIEWS=0 packs every bundle adjustment
    on the host as the reference does (ptz_incremental_optimizer.cc:420-440 -> ptzray_optimizer.cc:537-552, 799-850).  Same
    arrays, same initial rays: identical events (with LM iteration counts), registered sets and bit-identical cameras, solo and
    in lock step."""
    shapes = [(1, 20, True), (3, 24, False), (6, 40, True)]
    tables, cam0 = [], []
    for seed, n_views, bi in shapes:
        tb = pkg.synth.make_match_table(pkg.synth.make_scene(seed, n_views, 100), bidirectional=bi)
        tables.append(tb)
        c = np.zeros((tb.n_img, 15)); c[:, 0] = c[:, 1] = 1.0
        cam0.append(c)
    views = [pkg.hostlib.incremental_solve(tb, c, max_iter=200) for tb, c in zip(tables, cam0)]
    vbatch, vstats = pkg.hostlib.incremental_solve_batch(tables, cam0, max_iter=200)
    monkeypatch.setenv("PTZ_IBA_VIEWS", "0")
    packed = [pkg.hostlib.incremental_solve(tb, c, max_iter=200) for tb, c in zip(tables, cam0)]
    for a, b, c in zip(views, packed, vbatch):
        assert a["ok"] and b["ok"] and c["ok"]
        assert a["events"] == b["events"] == c["events"]
        assert a["registered"] == b["registered"] == c["registered"]
        assert np.array_equal(a["cameras"], b["cameras"]) and np.array_equal(a["cameras"], c["cameras"])
    assert vstats["ba_batches"] < vstats["ba_problems"]  # (the rigs' bundle adjustments of a round are one batch)
    # the registration attempts likewise: entries of a match table resident on the device (ptz_krt_table_create /
    # ptz_krt_solve_attempts; default) against their pixels packed for every launch (PTZ_IBA_MATCH_TABLE=0), solo and in lock step
    monkeypatch.delenv("PTZ_IBA_VIEWS", raising=False)
    monkeypatch.setenv("PTZ_IBA_MATCH_TABLE", "0")
    no_table = [pkg.hostlib.incremental_solve(tb, c, max_iter=200) for tb, c in zip(tables, cam0)]
    nbatch, _ = pkg.hostlib.incremental_solve_batch(tables, cam0, max_iter=200)
    monkeypatch.delenv("PTZ_IBA_MATCH_TABLE", raising=False)
    for a, b, c in zip(views, no_table, nbatch):
        assert b["ok"] and c["ok"]
        assert a["events"] == b["events"] == c["events"]
        assert np.array_equal(a["cameras"], b["cameras"]) and np.array_equal(a["cameras"], c["cameras"])
