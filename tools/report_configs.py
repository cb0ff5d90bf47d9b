#!/usr/bin/env python3
"""Per-configuration report on ONE MI355X (SURVEY.md 8(d) 'Reported numbers'): C1..C5 of BASELINE.json with LM iterations/s,
views (queries)/s, wall time, iterations to termination, termination histogram, parity against the CPU oracle where the
oracle finishes in seconds, and the CPU oracle timed beside it (all usable cores and one core; numeric-diff and analytic).
Writes one JSON document to stdout.  usage: python tools/report_configs.py [--quick]"""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def relrot(orc, cam):
    R = [orc.rodrigues(c[4:7]) for c in cam]
    return np.stack([r @ R[0].T for r in R])


def timed_solve(pkg, scenes, reps=2):
    b = pkg.api.BaBatch(scenes); b.set_state(); b.solve()
    best = None
    for _ in range(reps):
        t = time.perf_counter(); summ = b.solve(); dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    cams, rays = b.get_state()
    b.close()
    return summ, best, cams


def summarise(summ, dt, n_views):
    its = [s["num_lm_steps"] for s in summ]
    term = {}
    for s in summ:
        term[str(s["termination_type"])] = term.get(str(s["termination_type"]), 0) + 1
    return dict(scenes=len(summ), wall_ms=1e3 * dt, lm_iterations_per_s=sum(its) / dt, views_per_s=len(summ) * n_views / dt,
                lm_steps_min_mean_max=[int(min(its)), float(np.mean(its)), int(max(its))], termination_histogram=term)


def parity(pkg, orc, scene, cam, summ, threads):
    out = {}
    for name, mode in (("numeric", orc.JAC_NUMERIC), ("analytic", orc.JAC_ANALYTIC)):
        t = time.perf_counter()
        ocam, _, _, osumm, _ = orc.ba_solve(scene, jacobian_mode=mode, num_threads=threads)
        dt = time.perf_counter() - t
        out[name] = dict(iterations_equal=bool(summ["num_iterations"] == osumm["num_iterations"]),
                         focal_max_rel_error=rel(cam[:, 0], ocam[:, 0]),
                         relative_rotation_max_abs_error=float(np.abs(relrot(orc, cam) - relrot(orc, ocam)).max()),
                         final_cost_rel_error=abs(summ["final_cost"] - osumm["final_cost"]) / osumm["final_cost"],
                         cpu_lm_iterations_per_s=osumm["num_lm_steps"] / dt, cpu_threads=threads)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="smaller C4 / C5 (for a smoke run)")
    args = ap.parse_args()
    pkg = ge.load_package(); orc = ge.load_oracle(); orc.build()
    cores = orc.usable_cores()
    rep = {"device": "1 x MI355X", "host_cores_used": cores}

    # C1: 20 views x ~100 obs, single scene
    c1 = pkg.synth.make_scene(0, 20, 100)
    summ, dt, cams = timed_solve(pkg, [c1], reps=3)
    rep["C1"] = dict(workload="20 views x 100 obs/view, one scene", n_obs=c1.n_obs, **summarise(summ, dt, 20),
                     parity_vs_oracle=parity(pkg, orc, c1, cams[0], summ[0], cores))
    t = time.perf_counter(); _, _, _, o1, _ = orc.ba_solve(c1, jacobian_mode=orc.JAC_NUMERIC, num_threads=1); d1 = time.perf_counter() - t
    rep["C1"]["cpu_oracle_1_thread_lm_iterations_per_s"] = o1["num_lm_steps"] / d1

    # C2: 200 x 500, single scene
    c2 = pkg.synth.make_scene(0, 200, 500)
    summ, dt, cams = timed_solve(pkg, [c2], reps=3)
    rep["C2"] = dict(workload="200 views x 500 obs/view, one scene", n_obs=c2.n_obs, **summarise(summ, dt, 200),
                     parity_vs_oracle=parity(pkg, orc, c2, cams[0], summ[0], cores))
    t = time.perf_counter(); _, _, _, o1, _ = orc.ba_solve(c2, jacobian_mode=orc.JAC_NUMERIC, num_threads=1, max_num_iterations=4); d1 = time.perf_counter() - t
    rep["C2"]["cpu_oracle_1_thread_lm_iterations_per_s"] = o1["num_lm_steps"] / d1

    # C3: WorldCup14 stand-in, 4 scenes of 60 views x 300 obs, 1280 x 720, pan range 120 degrees
    c3 = [pkg.synth.make_scene(100 + s, 60, 300, width=1280, height=720, pan_range_deg=120.0) for s in range(4)]
    summ, dt, cams = timed_solve(pkg, c3, reps=3)
    rep["C3"] = dict(workload="4 scenes x 60 views x 300 obs/view, 1280x720, 120-degree pan (WorldCup14 stand-in)", **summarise(summ, dt, 60),
                     parity_vs_oracle=parity(pkg, orc, c3[0], cams[0], summ[0], cores))

    # C4: 1000 C2-shaped scenes (distinct seeds cycled), one batch on one GPU
    n4 = 64 if args.quick else 1000
    distinct = 8 if args.quick else 32
    base = [pkg.synth.make_scene(s, 200, 500) for s in range(distinct)]
    t = time.perf_counter()
    summ, dt, _ = timed_solve(pkg, [base[i % distinct] for i in range(n4)], reps=1)
    rep["C4"] = dict(workload=f"{n4} scenes x 200 views x 500 obs/view in one batch ({distinct} distinct seeds cycled)", **summarise(summ, dt, 200),
                     total_with_create_s=time.perf_counter() - t)

    # C5: relocalization queries, 128 matches each
    n5 = 10000 if args.quick else 100000
    rep["C5"] = {}
    for name, ft in (("F", 0), ("FDist", 1)):
        rb = pkg.synth.make_reloc_batch(n5, 128, seed_id=1, factor_type=ft)
        pkg.api.krt_solve_batch(rb)
        t = time.perf_counter(); cam_w, s5, acc, dev_ms = pkg.api.krt_solve_batch(rb); dt = time.perf_counter() - t
        its = sum(s["num_lm_steps"] for s in s5)
        rep["C5"][name] = dict(queries=n5, matches_per_query=128, accepted=int(np.sum(acc)), kernel_ms=dev_ms, wall_ms_with_pcie=1e3 * dt,
                               queries_per_s_kernel=n5 / (dev_ms * 1e-3), lm_iterations_per_s_kernel=its / (dev_ms * 1e-3),
                               focal_median_rel_error_vs_truth=float(np.median(np.abs(cam_w[acc == 1, 0] / rb.cam_gt[acc == 1, 0] - 1))))
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
