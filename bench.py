#!/usr/bin/env python3
"""bench.py -- PTZ-IBA / PTZ-Reloc throughput on MI355X, on BASELINE.json's own configurations.

BASELINE metric: "LM iterations/sec + views calibrated/sec, 200-view synthetic PTZ rig, 1/2/4/8 GPU".

  --config C4 (default)  configs[3]: synthetic 1000-scene batch, 200 views x 500 obs/view each, every scene its own seed.
                         One batch of 1000 scenes PER GPU (weak scaling: rank r solves seeds r*1000 .. r*1000+999); a "step" is
                         one pass of the hot path over that batch: every scene solved from its initial guess to Ceres-style
                         termination (ptz_ba_batch_solve).  The line also carries configs[1] (ONE 200 x 500 rig alone, `c2_single_rig`,
                         the configuration the 10 k it/s target is quoted on) with its per-pass critical path, configs[4]
                         (`c5_reloc`, a bounded sample of the relocalization queries) and the PTZ-IBA orchestration on one rig.
  --config C2            configs[1] as the headline: one rig per GPU, a step = one solve of it.
  --config C5            configs[4] as the headline: 100 000 relocalization queries x 128 matches per GPU, a step = one
                         ptz_krt_solve_batch_device launch over all of them (metric: LM iterations/s; queries/s beside it).

  --scaling weak|strong  weak (default): --scenes per GPU, rank r solves seeds r*scenes .. ; strong: --scenes in TOTAL (the literal
                         configs[3]: 1000 scenes over the node), dealt to the ranks in contiguous blocks.
The line starts with a small `headline` object (the C4 figure, the single-rig figure the 10 k it/s target is quoted on, queries/s,
views/s), and `config` repeats those scalars, so that a truncated copy of the line still shows them.

Inputs (observations, structure, initial state) are resident in HBM before the timed region.  N > 1: one process per GPU
(torch.distributed, backend nccl = RCCL), every rank owns its own scenes, no data-path collective; the timed region is
bracketed by barrier + torch.cuda.synchronize() and the MAX over ranks is reported; the result gather (15 doubles per view)
is timed separately (`gather_ms`).

The roofline block prices every kernel family with SURVEY.md section 8(d)'s ALGORITHMIC bytes / flops per unit of work
(DESIGN.md section 5): the implementation's own intermediates (the W = Jc^T Jr rows) are not algorithmic traffic.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 matrix; the in-repo guide lists no f64 figure
F64_VALU_PEAK_TFLOPS = 78.6  # datasheet FP64 vector (same rate as the matrix path on MI355X)


# ----------------------------------------------------------------------------------------------------------------- models
def structural_tile_ops(scene, nc, nb=64):
    """Tile operations the factorisation executes on a scene's reduced camera system: the library skips the 64 x 64 tiles that
    are zero by structure (camera co-visibility closed under the fill of a symbolic factorisation, ptz_ba.hip / DESIGN.md
    section 4).  Returns (tile updates, tile triangular solves, diagonal tiles).  Same construction as the host code."""
    import numpy as np
    n = nc * scene.n_cam
    nt = (n + 1 + nb - 1) // nb
    tile = (np.arange(scene.n_cam) * nc) // nb
    tile_hi = (np.arange(scene.n_cam) * nc + nc - 1) // nb
    m = np.zeros((nt, nt), dtype=bool)
    bounds = np.flatnonzero(np.diff(scene.obs_ray)) + 1
    starts = np.concatenate([[0], bounds]); ends = np.concatenate([bounds, [len(scene.obs_ray)]])
    for a, b in zip(starts, ends):
        cams = scene.obs_cam[a:b]
        ts = np.unique(np.concatenate([tile[cams], tile_hi[cams]]))
        m[np.ix_(ts, ts)] = True
    m[(n // nb):, :] = True  # the tile row that holds the right-hand side is full
    m = np.tril(m)
    upd = 0
    for k in range(nt):
        rows = [i for i in range(k + 1, nt) if m[i, k]]
        for x in rows:
            for y in rows:
                if y <= x:
                    m[x, y] = True
                    upd += 1
    trsm = int(np.tril(m, -1).sum())
    return upd, trsm, nt


def family_models(scene, nc):
    """SURVEY.md section 8(d) per-unit figures for one scene and one LM pass, per kernel family of the library's profile:
    bytes are ALGORITHMIC HBM bytes (what any implementation must move), flops the factorisation's arithmetic."""
    n = nc * scene.n_cam
    n_obs, n_ray = scene.n_obs, scene.n_ray
    s_bytes = 8.0 * n * (n + 1) / 2            # reduced camera system, written once (K2) and read once (K3)
    import numpy as np
    track_len = np.bincount(scene.obs_ray, minlength=n_ray).astype(np.float64)
    n_ent = float((track_len * (track_len - 1) / 2).sum())  # camera-pair entries: two observations of one track
    upd, trsm, nt = structural_tile_ops(scene, nc)
    nb = 64
    return {
        "linearize": dict(bound="hbm", bytes=16.0 * n_obs + 104.0 * n_ray, unit="relinearisation"),     # K1
        "eval": dict(bound="hbm", bytes=16.0 * n_obs + 96.0 * n_ray, unit="lm_step"),                     # K4
        "ray_prep": dict(bound="hbm", bytes=96.0 * n_ray, unit="lm_step"),
        "schur": dict(bound="hbm", bytes=s_bytes, unit="lm_step",                                         # K2: S written once
                      # the kernel's own FP64 vector work (ptz_ba_kernels.h k_schur_f), priced against the FP64 vector peak, because that,
                      # not HBM, is what the counters say bounds it (profiles/)
                      # (round 5, k_schur_f: ~330 flop per observation -- Jr0, Jc0, Q = Jr0 E'', N = Q Jr0^T, the sums, the 8-double row --
                      #  and ~160 per entry -- P' = Rji (x, y, 1), the reciprocal, K = Q Mr^T, K Graw_b, Graw_a^T (.), counted from the source)
                      valu_flops=330.0 * n_obs + 160.0 * n_ent),
        "chol_syrk": dict(bound="mfma", flops=upd * 2.0 * nb ** 3 + nt * nb ** 3 / 3.0, unit="lm_step",   # tile updates + diagonal tiles
                          flops_dense=n ** 3 / 3.0 + 2.0 * n * n),
        "chol_panel": dict(bound="mfma", flops=trsm * 1.0 * nb ** 3, unit="lm_step"),                     # tile triangular solves
        "chol_backsolve": dict(bound="hbm", bytes=s_bytes, unit="lm_step"),                               # L read once
    }


def roofline_table(prof, models, units, steps, traffic=None):
    """Per family: device ms per solve, launches, algorithmic work, achieved rate and fraction of the bounding peak.
    units[name] = executions of that family's unit of work by ALL scenes during the timed region."""
    out = {}
    for name, p in prof.items():
        if p["launches"] <= 0:
            continue
        row = {"ms_per_solve": round(p["ms"] / steps, 4), "launches_per_solve": p["launches"] / steps,
               "avg_launch_us": round(1e3 * p["ms"] / p["launches"], 2)}
        mdl = models.get(name)
        if mdl is not None:
            u = units[mdl["unit"]]
            if mdl["bound"] == "hbm":
                work = mdl["bytes"] * u
                ach = work / (p["ms"] * 1e-3) / 1e9
                row.update(bound="hbm", algorithmic_bytes_per_unit=mdl["bytes"], unit=mdl["unit"], achieved_GBps=round(ach, 2),
                           frac=round(ach / HBM_PEAK_GBS, 5))
                if "valu_flops" in mdl:  # a second fraction, against the FP64 vector peak; `bound` names the higher of the two
                    tfl = mdl["valu_flops"] * u / (p["ms"] * 1e-3) / 1e12
                    row.update(frac_hbm=row["frac"], achieved_TFLOPs_f64_valu=round(tfl, 3), frac_f64_valu=round(tfl / F64_VALU_PEAK_TFLOPS, 5),
                               valu_flops_per_unit=mdl["valu_flops"])
                    if row["frac_f64_valu"] > row["frac_hbm"]:
                        row["bound"] = "f64_valu"
            else:
                work = mdl["flops"] * u
                ach = work / (p["ms"] * 1e-3) / 1e12
                row.update(bound="mfma", flops_per_unit=mdl["flops"], unit=mdl["unit"], achieved_TFLOPs=round(ach, 3),
                           frac=round(ach / F64_MFMA_PEAK_TFLOPS, 5))
            if traffic and name in traffic:
                t = traffic[name]
                row["traffic_bytes_per_launch"] = t["hbm_bytes"]
                row["traffic_note"] = t.get("note", "")
                if mdl["bound"] == "hbm":
                    alg_per_launch = work / p["launches"]
                    row["traffic_ratio"] = round(t["hbm_bytes"] / max(alg_per_launch, 1.0), 2)
        out[name] = row
    return out


def dominant_roofline(table, prof, models, units, traffic=None):
    cand = [k for k in table if k in models]
    dom = max(cand, key=lambda k: prof[k]["ms"])
    row, mdl, p = table[dom], models[dom], prof[dom]
    u = units[mdl["unit"]]
    if mdl["bound"] == "hbm":
        per_launch = mdl["bytes"] * u / p["launches"]
        ach = per_launch / (p["ms"] / p["launches"] * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                    algorithmic_bytes_per_launch=per_launch)
    else:
        per_launch = mdl["flops"] * u / p["launches"]
        ach = per_launch / (p["ms"] / p["launches"] * 1e-3) / 1e12
        roof = dict(bound="mfma", kernel=dom, achieved=ach, peak=F64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=ach / F64_MFMA_PEAK_TFLOPS, algorithmic_flops_per_launch=per_launch)
    roof.update(traffic=row.get("traffic_bytes_per_launch"), avg_launch_ms=p["ms"] / p["launches"], launches=p["launches"])
    if "frac_f64_valu" in row:  # (the contract's `bound` stays the roofline SURVEY 8(d) prices the family against; what the kernel is really bound by beside it)
        roof.update(bound_measured=row["bound"], frac_f64_valu=row["frac_f64_valu"], achieved_TFLOPs_f64_valu=row["achieved_TFLOPs_f64_valu"])
    if "traffic_ratio" in row:
        roof["traffic_ratio"] = row["traffic_ratio"]
        roof["traffic_source"] = row.get("traffic_note", "")
    return roof


def load_traffic(tag):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (same workload shape; PMC cannot be
    collected from inside the benchmark process)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    except (OSError, ValueError):
        return None
    if tj.get("workload_tag") != tag:
        return None
    return tj.get("families")


def c2_floor_model():
    """What ONE LM iteration of the C2 rig costs at least with the present structure, term by term (profiles/NOTES_r06.md section 1:
    every constant is a probe's or the chain timeline's measurement on MI355X at the ~2.45 GHz the shader clock holds during the chain,
    none is fitted to the bench figure; on another part or clock they go stale).  The 800 x 800 reduced camera system is 13 tiles of 64
    columns in 8 levels of the dissected elimination order; a level is one diagonal tile's factorisation (four 16-pivot sweeps of one
    wave, a rank-16 update between two sweeps) plus the hand-over to the next level's tile on another compute unit."""
    levels, sweeps, pivots = 8, 4, 16
    ns_pivot = 165 / 2.45                # 165 cycles per pivot of the software-pipelined DPP sweep (sweep16_probe; issue-bound: issue_probe)
    t = {
        "chain_pivots": levels * sweeps * pivots * ns_pivot * 1e-3,      # the dependent chain itself
        "chain_sweep_lds": levels * sweeps * 0.45,                       # a sweep's column block from LDS and back (1 100 cycles)
        "chain_rank16_updates": levels * (sweeps - 1) * 0.40,            # MFMA update + two barriers between two sweeps
        "chain_tile_in_and_out_of_lds": levels * 1.3,                    # accumulators -> image, last block's publication
        "chain_hand_overs": 5 * 2.2,                                     # chased columns: sc1 flag + block fetch, last solve round, a quarter of the update
        "chain_two_producer_levels": 2 * 6.5,                            # levels 2 and 4: two producers end together -- taken side by side, then the second's 64 update MFMAs
        "chain_late_off_diagonal_tile": 5.0,                             # tile (9, 8): six columns of two solves + an update each pace it (FP64 MFMA: 64 cycles each)
        "chain_last_tile_inverse": 3.0,
        "back_substitution": 45 * 32768 / 142e9 * 1e6 + 9 * 0.9 + 2.5,   # 45 of the 76 tiles through a compute unit at 142 GB/s (the two arcs on a workgroup each), nine inverse groups, start
        "k_eval": 14.0, "k_lin_cam_and_control": 11.0, "k_schur_f": 14.0, "k_ray_prep": 5.0,   # longest thread's work, one wave per SIMD (k_eval: four lanes per ray)
        "launch_boundaries": 6 * 1.6,                                    # six dependent kernels per pass, enqueued eagerly
    }
    out = {k: round(v, 1) for k, v in t.items()}
    out["total_us"] = round(sum(t.values()), 1)
    out["measured_on"] = "MI355X (gfx950), shader clock ~2.45 GHz during the chain kernel; round-6 probes (tools/probes/hip/issue_probe.hip, sweep16_probe.hip, probe_r6_tl.sh)"
    out["note"] = ("sum of the dependent stretches of one pass.  The north star's 10 k it/s is 100 us per iteration: below this "
                   "structure's floor -- NOTES_r06 section 1 says what the round removed (L2 write-backs and invalidates of the hand-overs) and what is left")
    return out


# ------------------------------------------------------------------------------------------------------------------ legs
def single_rig_leg(pkg, scene, device_id):
    """BASELINE configs[1]: ONE 200 x 500 rig alone on the GPU.  Wall time of a whole solve (the library's defaults: for one rig the
    passes are enqueued as they are, nothing profiled), then the same solve with HIP events around every kernel family: the per-pass
    critical path in microseconds."""
    import torch
    b1 = pkg.api.BaBatch([scene], device_id=device_id)
    b1.set_state(); b1.solve()
    best = None
    for _ in range(5):
        t1 = time.perf_counter(); s1 = b1.solve(); torch.cuda.synchronize(); d1 = time.perf_counter() - t1
        best = d1 if best is None else min(best, d1)
    steps = s1[0]["num_lm_steps"]
    b1.set_profiling(True); b1.solve(); prof = b1.get_profile(); b1.set_profiling(False)
    dev_ms = b1.last_solve_ms()
    b1.close()
    per_pass = {k: round(1e3 * v["ms"] / steps, 2) for k, v in prof.items() if v["launches"] > 0}
    us = 1e6 * best / steps
    floor = c2_floor_model()
    floor.update(measured_us_per_lm_iteration=round(us, 1), measured_over_floor=round(us / floor["total_us"], 2))
    return {"workload": f"C2: one rig, {scene.n_cam} views x {scene.n_obs // scene.n_cam} obs/view ({scene.n_obs} observations, "
                        f"{scene.n_ray} rays), solved alone",
            "lm_iterations_per_s": steps / best, "views_per_s": scene.n_cam / best, "ms_per_solve": 1e3 * best, "lm_steps": steps,
            "us_per_lm_iteration": 1e6 * best / steps,
            "per_pass_critical_path_us": per_pass,
            "floor_model": floor,
            "per_pass_critical_path_note": "HIP events around every kernel family of an eagerly enqueued solve (event pairs add "
                                           f"a few us per family); that profiled solve took {dev_ms:.2f} ms on the device",
            "termination_type": s1[0]["termination_type"]}


class RelocRun:
    """BASELINE configs[4]: relocalization queries x 128 matches, queries resident in HBM, one launch for all of them."""

    def __init__(self, pkg, rb, device_id):
        import numpy as np
        import torch
        self.pkg, self.rb, self.device_id = pkg, rb, device_id
        self.dev = torch.device("cuda", device_id)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)  # noqa: E731
        self.d_ptr, self.d_ref, self.d_cur, self.d_cref = t(rb.match_ptr), t(rb.uv_ref), t(rb.uv_cur), t(rb.cam_ref)
        self.init = t(rb.cam_init)
        self.d_ccur = self.init.clone()
        self.d_sum = torch.zeros((rb.n_query, 64), dtype=torch.uint8, device=self.dev)  # sizeof(ptz_lm_summary) = 64
        self.d_acc = torch.zeros(rb.n_query, dtype=torch.int32, device=self.dev)
        self.launch()
        torch.cuda.synchronize()

    def launch(self):
        import torch
        st = torch.cuda.current_stream(self.dev)
        self.pkg.api.krt_solve_batch_device(self.rb.n_query, self.d_ptr, self.d_ref, self.d_cur, self.d_cref, self.d_ccur, self.d_sum,
                                            self.d_acc, factor_type=self.rb.factor_type, stream=st.cuda_stream)

    def run(self, steps):
        """`steps` launches, each from the initial cameras.  Returns (wall seconds for all of them, leg dict)."""
        import numpy as np
        import torch
        st = torch.cuda.current_stream(self.dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            self.d_ccur.copy_(self.init)
            ev[2 * k].record(st)
            self.launch()
            ev[2 * k + 1].record(st)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        kern_ms = sum(ev[2 * k].elapsed_time(ev[2 * k + 1]) for k in range(steps)) / steps
        n_query = self.rb.n_query
        summ = np.frombuffer(self.d_sum.cpu().numpy().tobytes(), dtype=np.int32).reshape(n_query, 16)
        lm_steps = int(summ[:, 2].sum())
        acc = int(self.d_acc.sum().item())
        # 2 x 16 B x M per LM iteration (linearise + candidate) + 240 B of state; ~300 flop per match and iteration
        alg_bytes = 2.0 * 16.0 * 128 * lm_steps + 240.0 * lm_steps
        alg_flops = 300.0 * 128 * lm_steps
        gbps = alg_bytes / (kern_ms * 1e-3) / 1e9
        tfl = alg_flops / (kern_ms * 1e-3) / 1e12
        leg = {"workload": f"C5: {n_query} relocalization queries x 128 matches ({'FDist' if self.rb.factor_type else 'F'} factor), resident in "
                           "HBM, one ptz_krt_solve_batch_device launch", "queries": n_query, "matches": int(self.rb.match_ptr[-1]),
               "queries_per_s": n_query / (kern_ms * 1e-3), "lm_iterations_per_s": lm_steps / (kern_ms * 1e-3),
               "kernel_ms": kern_ms, "wall_ms_per_launch": 1e3 * wall / steps, "lm_iterations": lm_steps, "accepted": acc,
               "roofline": {"kernel": "k_krt", "bound": "hbm" if gbps / HBM_PEAK_GBS > tfl / F64_VALU_PEAK_TFLOPS else "f64_valu",
                            "achieved_GBps": gbps, "frac_hbm": gbps / HBM_PEAK_GBS, "achieved_TFLOPs_f64_valu": tfl,
                            "frac_f64_valu": tfl / F64_VALU_PEAK_TFLOPS, "algorithmic_bytes_per_launch": alg_bytes,
                            "algorithmic_flops_per_launch": alg_flops,
                            "note": "bytes: 2 x 16 B x M + 240 B per LM iteration, flops: ~300 per match and iteration (SURVEY 8(d)); "
                                    "`bound` names the peak the kernel sits closer to"}}
        return wall, leg, lm_steps


def iba_leg(pkg, scene):
    """Second half of BASELINE's metric: views calibrated / s of the full incremental pipeline (PtzIncrementalOptimizer, C++
    host class, every solve on the device): one rig of the same shape, starting from uncalibrated cameras."""
    import numpy as np
    tb = pkg.synth.make_match_table(scene)
    cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
    pkg.hostlib.incremental_solve(tb, cam0, max_iter=200)  # warm-up (resource pool, code objects)
    t1 = time.perf_counter(); r = pkg.hostlib.incremental_solve(tb, cam0, max_iter=200); d1 = time.perf_counter() - t1
    reg = r["registered"]
    solve_ms = float(r["timing_ms"]["construct"] + r["timing_ms"]["solve"])
    return {"views": tb.n_img, "registered": len(reg), "wall_ms": 1e3 * d1, "views_per_s": len(reg) / d1,
            "solve_ms": solve_ms, "views_per_s_solve_only": len(reg) / (solve_ms * 1e-3),
            "bundle_adjustments": sum(1 for e in r["events"] if e[0] == 2), "lm_iterations": r["lm_iterations"],
            "registrations": sum(1 for e in r["events"] if e[0] == 1),
            "max_focal_rel_error": float(np.abs(r["cameras"][reg, 0] / scene.cam_gt[reg, 0] - 1).max()) if reg else None,
            "timing_ms": {k: round(float(v), 2) for k, v in r["timing_ms"].items()}}


def iba_batch_leg(pkg, scenes, tables, device_id):
    """Views calibrated / s as THROUGHPUT: N rigs through the full incremental pipeline in lock step (PtzIncrementalOptimizer::
    SolveBatch, host/device_batcher.h) -- every round's bundle adjustments in one ptz_ba_batch, its registration attempts in one
    ptz_krt_solve_batch launch; decisions identical to N solo runs (tests/test_gpu_configs.py).  The reference calibrates its
    scenes one after the other (run_ptzba_synthetic.sh:4-13)."""
    import numpy as np
    cam0 = []
    for tb in tables:
        c = np.zeros((tb.n_img, 15)); c[:, 0] = c[:, 1] = 1.0
        cam0.append(c)
    pkg.hostlib.incremental_solve_batch(tables, cam0, max_iter=200, device_id=device_id)  # warm-up at full size (resource pool: pinned and device blocks of every size class the rounds ask for)
    t1 = time.perf_counter()
    res, st = pkg.hostlib.incremental_solve_batch(tables, cam0, max_iter=200, device_id=device_id, events_as_array=True)
    d1 = time.perf_counter() - t1
    reg = sum(len(r["registered"]) for r in res)
    ferr = [float(np.abs(r["cameras"][r["registered"], 0] / sc.cam_gt[r["registered"], 0] - 1).max()) for r, sc in zip(res, scenes) if r["registered"]]
    return {"rigs": len(tables), "views": sum(tb.n_img for tb in tables), "registered": reg, "solved_rigs": sum(1 for r in res if r["ok"]),
            "wall_ms": 1e3 * d1, "views_per_s": reg / d1, "solve_ms": st["wall_ms"], "views_per_s_solve_only": reg / (1e-3 * st["wall_ms"]),
            "lm_iterations": sum(r["lm_iterations"] for r in res),
            "rounds": st["rounds"], "bundle_adjustments": st["ba_problems"], "bundle_adjustment_batches": st["ba_batches"],
            "registration_launches": st["krt_launches"], "registration_attempts": st["krt_queries"],
            "ms_in_batched_bundle_adjustments": st["ba_ms"], "ms_in_registration_launches": st["krt_ms"],
            "max_focal_rel_error": max(ferr) if ferr else None}


def parity_leg(pkg):
    """SURVEY 8(d) "Reported numbers": the parity figures of the smoke-sized check (BASELINE configs[0]: 20 views x ~100 obs), device
    against the oracle in its reference-faithful numeric-differentiation mode, next to the throughput they belong to."""
    import numpy as np
    orc = ge.load_oracle()
    orc.build()
    sc = pkg.synth.make_scene(0, 20, 100)
    cam, _, summ = pkg.api.ba_solve(sc)
    ocam, _, _, osumm, _ = orc.ba_solve(sc, jacobian_mode=orc.JAC_NUMERIC)
    R = [orc.rodrigues(c[4:7]) for c in cam]
    Ro = [orc.rodrigues(c[4:7]) for c in ocam]
    rot = max(float(np.abs(a @ R[0].T - b @ Ro[0].T).max()) for a, b in zip(R, Ro))
    return {"workload": "C1 (BASELINE configs[0]): 20 views x ~100 obs/view, device vs numeric-diff oracle",
            "max_rel_f": float(np.abs(cam[:, 0] - ocam[:, 0]).max() / np.abs(ocam[:, 0]).max()), "max_rot": rot,
            "rel_final_cost": abs(summ["final_cost"] - osumm["final_cost"]) / osumm["final_cost"],
            "iteration_count_equal": bool(summ["num_iterations"] == osumm["num_iterations"]),
            "termination_equal": bool(summ["termination_type"] == osumm["termination_type"]),
            "iterations": int(summ["num_iterations"]), "tolerance": 1e-6}


def cpu_baseline_leg(scenes, budget_s=12.0):
    """The reference-faithful CPU oracle ("port": central-difference Jacobians over all 18 block parameters as
    ceres::NumericDiffCostFunction, Ceres-1.14 LM policy, dense Schur) on scenes of the same workload: all usable host
    cores (the figure of record), then one scene in closed-form-Jacobian mode and a few iterations on ONE thread, to
    separate the algorithmic from the hardware speed-up.  A restatement, not the Ceres/OpenCV binary."""
    orc = ge.load_oracle()
    orc.build()
    cores = orc.usable_cores()
    n_done, steps_done, t_cpu = 0, 0, 0.0
    rates = []  # LM iterations / s of every solve: SURVEY 8(d) asks for the median of >= 5 runs
    for sc in scenes:
        t1 = time.perf_counter()
        _, _, _, osumm, _ = orc.ba_solve(sc, jacobian_mode=orc.JAC_NUMERIC, num_threads=cores)
        dt = time.perf_counter() - t1
        t_cpu += dt
        n_done += 1
        steps_done += osumm["num_lm_steps"]
        rates.append(osumm["num_lm_steps"] / dt)
        if t_cpu > budget_s and n_done >= 5:
            break
        if t_cpu > 2.5 * budget_s:
            break
    import statistics
    t1 = time.perf_counter()
    _, _, _, asumm, _ = orc.ba_solve(scenes[0], jacobian_mode=orc.JAC_ANALYTIC, num_threads=cores)
    t_an = time.perf_counter() - t1
    t1 = time.perf_counter()
    _, _, _, ssumm, _ = orc.ba_solve(scenes[0], jacobian_mode=orc.JAC_NUMERIC, num_threads=1, max_num_iterations=6)
    t_1 = time.perf_counter() - t1
    return {"value": statistics.median(rates), "unit": "LM iterations/s", "cores": cores, "kind": "port",
            "value_mean": steps_done / t_cpu, "runs": n_done, "runs_lm_iterations_per_s": [round(r, 2) for r in rates],
            "sample": f"median over {n_done} scenes of the same workload (seeds {scenes[0].seed:#x}..), each solved to termination: "
                      f"{steps_done} LM iterations in {t_cpu:.2f} s in all, numeric-diff oracle with OpenMP on {cores} cores",
            "analytic_jacobians_all_cores": {"value": asumm["num_lm_steps"] / t_an, "sample": f"1 scene, {asumm['num_lm_steps']} LM iterations in {t_an:.2f} s"},
            "numeric_one_thread": {"value": ssumm["num_lm_steps"] / t_1, "cores": 1,
                                   "sample": f"1 scene, first {ssumm['num_lm_steps']} LM iterations in {t_1:.2f} s"}}


def reloc_cpu_baseline_leg(pkg, n_sample=6000, budget_s=10.0):
    """The relocalization loop of the reference (run_ptz_reloc.cc:68-118: one KRTOptimizer per query, numeric differentiation
    over all 15 parameters, DENSE_QR) as the oracle restates it, on a bounded sample of the same queries: whole queries dealt
    to all usable host cores, and on one thread.  A port, not the Ceres/OpenCV binary."""
    orc = ge.load_oracle()
    orc.build()
    cores = orc.usable_cores()
    rb = pkg.synth.make_reloc_queries(n_sample, 128, seed_id=1, factor_type=0)
    out = {}
    for label, nt, n in (("all_cores", cores, n_sample), ("one_thread", 1, max(200, n_sample // 8))):
        t1 = time.perf_counter()
        _, summ, acc = orc.krt_solve_batch(rb, n_query=n, num_threads=nt, jacobian_mode=orc.JAC_NUMERIC)
        dt = time.perf_counter() - t1
        its = sum(s_["num_lm_steps"] for s_ in summ)
        out[label] = {"queries_per_s": n / dt, "lm_iterations_per_s": its / dt, "cores": nt,
                      "sample": f"{n} queries x 128 matches (the first of the same seeded stream), {its} LM iterations in {dt:.2f} s, accepted {int(acc.sum())}"}
        if dt > budget_s:
            break
    a = out["all_cores"]
    return {"value": a["lm_iterations_per_s"], "unit": "LM iterations/s", "queries_per_s": a["queries_per_s"], "cores": a["cores"], "kind": "port",
            "sample": a["sample"] + f"; numeric-diff + QR oracle, whole queries over {a['cores']} threads", "one_thread": out.get("one_thread")}


# ------------------------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["C4", "C2", "C5"], default="C4")
    ap.add_argument("--scenes", type=int, default=None, help="scenes per GPU (C4: 1000, C2: 1)")
    ap.add_argument("--batch", type=int, default=None, help="alias of --scenes")
    ap.add_argument("--distinct", type=int, default=None, help="distinct seeded scenes per GPU (default: all of them)")
    ap.add_argument("--queries", type=int, default=100000, help="C5: queries per GPU")
    ap.add_argument("--views", type=int, default=200)
    ap.add_argument("--obs", type=int, default=500)
    ap.add_argument("--workers", type=int, default=None, help="host processes that generate the synthetic scenes")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --scenes per GPU; strong: --scenes in total over all GPUs (BASELINE configs[3] literally: 1000 scenes, 8 GPUs)")
    ap.add_argument("--scene-cache", default=os.environ.get("PTZ_SCENE_CACHE", os.path.join("/tmp", f"ptz_scene_cache_{os.getuid()}")),
                    help="directory for generated scenes (.npz, keyed by seed and shape; '' disables)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--iba-rigs", type=int, default=64, help="rigs of the batched PTZ-IBA leg (0 skips it)")
    ap.add_argument("--dump-gathered", default=None,
                    help="rank 0 writes the gathered result blocks (one row per scene of the whole job: 15 doubles per view, then termination, "
                         "iterations, final cost) to this .npy file -- what the multi-rank test compares with solo solves")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed steps: no single-rig / reloc / orchestration / CPU legs, so that a rocprofv3 --stats "
                         "summary of this command averages over the timed launch shape alone")
    args = ap.parse_args()

    import numpy as np

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    pkg = ge.load_package()

    # ---- synthetic inputs first: the generator forks worker processes, which must happen before this process touches the GPU
    n_arg = args.scenes or args.batch or (1000 if args.config == "C4" else 1)
    if args.scaling == "strong":  # n_arg scenes in total, contiguous blocks per rank
        rg = pkg.sharding.shard_range(n_arg, rank, world)
        first_scene, n_scenes, n_total = rg.start, len(rg), n_arg
    else:
        first_scene, n_scenes, n_total = rank * n_arg, n_arg, n_arg * world
    t_gen = time.perf_counter()
    scenes, base = [], []
    if args.config in ("C4", "C2"):
        distinct = max(1, min(n_scenes, args.distinct or n_scenes))
        workers = args.workers
        if workers is None and world > 1:
            workers = max(1, min(16, pkg.synth.usable_cores() // world))  # (the cgroup quota, not the logical CPU count: eight ranks share it)
        base = pkg.synth.make_scenes([first_scene + i for i in range(distinct)], args.views, args.obs, workers=workers,
                                     cache_dir=args.scene_cache or None)
        scenes = [base[i % distinct] for i in range(n_scenes)]
    rb_c5 = pkg.synth.make_reloc_queries(args.queries, 128, seed_id=1 + rank, factor_type=0) if args.config == "C5" else None
    extras = not args.headline_only
    c2_scene = None
    iba_scenes, iba_tables = [], []
    if extras and rank == 0:
        c2_scene = base[0] if base else pkg.synth.make_scene(0, args.views, args.obs)
        if args.config != "C5" and args.iba_rigs > 0:
            iba_scenes = (base[: args.iba_rigs] if len(base) >= args.iba_rigs else
                          pkg.synth.make_scenes(range(args.iba_rigs), args.views, args.obs, cache_dir=args.scene_cache or None))
            iba_tables = pkg.synth.make_match_tables(iba_scenes)
    t_gen = time.perf_counter() - t_gen

    import torch
    dist = None
    # Debug aid for boxes with ONE GPU: PTZ_BENCH_SHARED_GPU=1 lets several ranks share device 0 (gloo for the collectives),
    # which exercises the multi-rank bookkeeping of this script; it is not a measurement mode.
    shared_gpu = os.environ.get("PTZ_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    backend = None
    # under torch.distributed.run (RANK / MASTER_ADDR in the environment) the collectives are set up even for one rank, so that
    # `--gpus 1` launched the way the driver launches N > 1 exercises the same RCCL code path
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist
        backend = "gloo" if shared_gpu else "nccl"
        if shared_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if shared_gpu else torch.device("cuda", local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = {}
    gather_ms = None
    if args.config == "C5":
        rr = RelocRun(pkg, rb_c5, local_rank)
        rr.run(max(args.warmup, 1))
        barrier()
        t0 = time.perf_counter()
        _, leg, lm_steps_launch = rr.run(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        lm_steps = lm_steps_launch * args.steps
        units_done = float(args.queries * args.steps)
        summ = []
    else:
        batch = pkg.api.BaBatch(scenes, device_id=local_rank)
        batch.set_state()  # observations, structure and the initial state are now resident in HBM
        for _ in range(args.warmup):
            summ = batch.solve()
        batch.set_profiling(True)  # HIP events around every kernel family on the solver's own stream
        barrier()
        t0 = time.perf_counter()
        lm_steps = 0
        jac_evals = 0
        step_ms = []
        for _ in range(args.steps):
            ts = time.perf_counter()
            summ = batch.solve()  # (returns when the batch's streams have drained)
            step_ms.append(1e3 * (time.perf_counter() - ts))
            lm_steps += sum(s["num_lm_steps"] for s in summ)
            jac_evals += sum(s["num_jacobian_evals"] + 1 for s in summ)  # + the re-evaluation after the Jacobi scales are fixed
        barrier()
        elapsed = time.perf_counter() - t0
        prof = batch.get_profile()
        batch.set_profiling(False)
        units_done = float(n_scenes * args.views * args.steps)
        # results to rank 0: 15 doubles per view and a summary triple per scene (the only collective of the path)
        if dist is not None:
            cams, _ = batch.get_state()
            payload = np.stack([np.concatenate([c.reshape(-1), [s["termination_type"], s["num_iterations"], s["final_cost"]]])
                                for c, s in zip(cams, summ)])
            barrier()
            tg = time.perf_counter()
            gathered = pkg.sharding.gather_results(list(range(first_scene, first_scene + n_scenes)), payload, n_total, dist,
                                                   None if shared_gpu else dev)
            barrier()
            gather_ms = 1e3 * (time.perf_counter() - tg)
            assert gathered.shape[0] == n_total
            if rank == 0 and args.dump_gathered:
                np.save(args.dump_gathered, gathered)

    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ws = torch.tensor([float(lm_steps), units_done, float(sum(1 for s in summ if s["termination_type"] == 0))], dtype=torch.float64, device=dev)
    per_rank = None
    if dist is not None:
        parts = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)
        per_rank = [1e3 * float(p.item()) / args.steps for p in parts]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ws, op=dist.ReduceOp.SUM)
    t_max = float(tt.item())
    total_steps, total_units, n_conv = (float(v) for v in ws.tolist())

    if rank == 0:
        par = {"world_size": world, "backend": backend or "none", "ranks_ms_per_step": per_rank, "gather_ms": gather_ms,
               "world_size_seen_by_collective": (dist.get_world_size() if dist is not None else 1),
               "first_scene_of_rank0": first_scene, "scenes_of_rank0": n_scenes}
        body = {}  # everything behind the contract keys, in the order it is printed
        if args.config == "C5":
            config = {"workload": f"C5 (BASELINE configs[4]): {args.queries} relocalization queries x 128 matches per GPU, F factor, "
                                  "queries resident in HBM", "queries_per_gpu": args.queries, "parallelism": f"query-sharded x{world}"}
            body["queries_per_s"] = total_units / t_max
            roof = {"bound": "hbm", "kernel": "k_krt", "achieved": leg["roofline"]["achieved_GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": leg["roofline"]["frac_hbm"], "traffic": None,
                    "avg_launch_ms": leg["kernel_ms"], "detail": leg["roofline"]}
            body["c5_reloc"] = leg
        else:
            name = "C4 (BASELINE configs[3])" if args.config == "C4" and n_arg == 1000 else ("C2 (BASELINE configs[1])" if n_arg == 1 else "C4-shaped")
            config = {"workload": f"{name}: {n_scenes} synthetic scenes on rank 0 ({n_total} over {world} GPU{'s' if world > 1 else ''}, {args.scaling} "
                                  f"scaling), one batch per GPU, each scene {args.views} views x {args.obs} obs/view (PTZRay factor), every scene its "
                                  "own seed, solved from its initial guess to termination",
                      "scenes_per_gpu": n_scenes, "scenes_total": n_total, "distinct_seeds_per_gpu": len(base), "views": args.views,
                      "obs_per_view": args.obs, "n_obs_per_scene_mean": float(np.mean([s.n_obs for s in base])),
                      "n_ray_per_scene_mean": float(np.mean([s.n_ray for s in base])),
                      "parallelism": f"scene-sharded x{world}", "scene_generation_s": round(t_gen, 1)}
            body["views_per_s"] = total_units / t_max
            body["lm_steps_per_solve"] = total_steps / args.steps / world
            body["converged_scenes"] = int(n_conv)
            body["scenes_total"] = n_total
            models = family_models(base[0], batch.nc)
            units = {"lm_step": float(lm_steps), "relinearisation": float(jac_evals)}
            traffic = load_traffic(f"{args.config}:{n_scenes}x{args.views}x{args.obs}")
            table = roofline_table(prof, models, units, args.steps, traffic)
            roof = dominant_roofline(table, prof, models, units, traffic)
            body["kernel_families"] = table
            body["roofline_note"] = ("achieved = SURVEY 8(d) algorithmic bytes (flops) of the family x units executed by all scenes in the "
                                     "timed region / family device time (HIP events on the solver's stream); traffic = rocprofv3 PMC bytes "
                                     "per launch (profiles/).  Since round 3 the RAY side of an accepted step's linearisation (V, g_r) is computed by the second pass of "
                                     "k_eval and only its camera side by the `linearize` family: the `linearize` and `eval` rows share SURVEY 8(d)'s K1 + K4 bytes and "
                                     "are best read together")
        if args.config != "C5":
            body["ms_per_step_median"] = float(np.median(step_ms))
            body["ms_per_step_each"] = [round(v, 3) for v in step_ms]
            body["value_at_median_step"] = total_steps / args.steps / (1e-3 * float(np.median(step_ms)))  # (rank 0's step times)
        body["parallel"] = par
        if extras and args.config != "C5":
            # The timed region above runs with per-family profiling, which makes the library enqueue eagerly and solve the batch as
            # one scene group (exclusive kernel timings for the roofline).  The library's default on the same resident batch:
            t2 = time.perf_counter()
            s2 = batch.solve()
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t2
            body["default_pipeline"] = {"lm_iterations_per_s": sum(s["num_lm_steps"] for s in s2) / d2, "ms_per_solve": 1e3 * d2,
                                        "note": "library defaults (graph-replayed passes, scene groups as the library chooses), rank 0, profiling off"}
        if extras:
            try:
                rd, cp = pkg.api.hbm_bandwidth(local_rank)
                body["measured_peaks"] = {"hbm_read_GBps": rd, "hbm_copy_GBps": cp, "mfma_f64_TFLOPs": pkg.api.mfma_f64_peak(local_rank),
                                          "note": "streaming read / copy of 4 GB; register-resident v_mfma_f64_16x16x4_f64 loop"}
            except pkg.api.PtzError:
                pass
    if args.config != "C5":
        batch.close()
    if rank == 0:
        cpu = None
        legs_failed = []  # side legs that raised: reported at the top level of the line (a regression in a leg must not pass unnoticed)
        def side_leg(name, fn):
            # the legs beside the headline never take the line down with them: a leg that fails is reported as such, with its error
            try:
                body[name] = fn()
            except Exception as e:  # noqa: BLE001 -- whatever it was, the headline measured above stands
                body[name] = {"error": f"{type(e).__name__}: {e}"}
                legs_failed.append(name)
                print(f"[bench] leg {name} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)

        if extras:
            pkg.api.trim_cache()
            side_leg("c2_single_rig", lambda: single_rig_leg(pkg, c2_scene, local_rank))
            if args.config != "C5":
                def c5_leg():
                    r = RelocRun(pkg, pkg.synth.make_reloc_queries(args.queries, 128, seed_id=1, factor_type=0), local_rank).run(5)[1]  # configs[4] at full size
                    r["note"] = "configs[4] at full size (--queries per GPU); `bench.py --config C5` makes it the headline"
                    return r
                side_leg("c5_reloc", c5_leg)
            side_leg("ptz_iba", lambda: iba_leg(pkg, c2_scene))
            if iba_tables:
                side_leg("ptz_iba_batch", lambda: iba_batch_leg(pkg, iba_scenes, iba_tables, local_rank))
        if extras:
            side_leg("parity", lambda: parity_leg(pkg))
        if world == 1 and extras and not args.no_cpu_baseline:
            if "error" not in body.get("c5_reloc", {"error": 1}):
                try:
                    body["c5_reloc"]["cpu_baseline"] = reloc_cpu_baseline_leg(pkg)
                except Exception as e:  # noqa: BLE001
                    body["c5_reloc"]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
            if args.config == "C5":
                cpu = body.get("c5_reloc", {}).get("cpu_baseline")
                if cpu is not None and "error" in cpu:
                    cpu = None
            else:
                cpu = cpu_baseline_leg(base if base else [c2_scene])
        value = total_steps / t_max
        # the figures a reader looks for first, at the front of the line and again as scalars of `config`
        headline = {"c4_lm_iterations_per_s" if args.config == "C4" else "lm_iterations_per_s": value}
        if "default_pipeline" in body:
            headline["default_pipeline_lm_iterations_per_s"] = body["default_pipeline"]["lm_iterations_per_s"]
        if "error" not in body.get("c2_single_rig", {"error": 1}):
            headline["c2_single_rig_lm_iterations_per_s"] = body["c2_single_rig"]["lm_iterations_per_s"]
            headline["c2_single_rig_us_per_lm_iteration"] = body["c2_single_rig"]["us_per_lm_iteration"]
            headline["c2_target_lm_iterations_per_s"] = 10000.0
        if "error" not in body.get("c5_reloc", {"error": 1}):
            headline["c5_queries_per_s"] = body["c5_reloc"]["queries_per_s"]
        if "error" not in body.get("ptz_iba", {"error": 1}):
            headline["ptz_iba_views_per_s"] = body["ptz_iba"]["views_per_s"]
        if "error" not in body.get("ptz_iba_batch", {"error": 1}):
            headline["ptz_iba_batch_views_per_s"] = body["ptz_iba_batch"]["views_per_s"]
            headline["ptz_iba_batch_views_per_s_inside_the_call"] = body["ptz_iba_batch"]["views_per_s_solve_only"]
        if cpu is not None:
            headline["cpu_port_lm_iterations_per_s"] = cpu["value"]
        config.update({k: (round(v, 1) if isinstance(v, float) else v) for k, v in headline.items()})
        out = {
            "headline": headline,
            "metric": "LM iterations/sec (PTZ-IBA global BA, 200-view synthetic PTZ rig)" if args.config != "C5"
                      else "LM iterations/sec (PTZ-Reloc single-view LM, 128 matches per query)",
            "value": value,
            "unit": "LM iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            # BASELINE.md publishes no number for this metric, so the contract's ratio is null; the ratio to the CPU port of the
            # reference algorithm timed on this box's host cores (cpu_baseline below) is reported beside it under its own name
            "vs_baseline": None,
            "vs_cpu_baseline": (value / cpu["value"]) if cpu else None,
            "dtype": "f64",
            "data": "synthetic",
            "config": config,
            "roofline": roof,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if "default_pipeline" in body:  # what a caller of the library gets (graph-replayed passes, two scene groups): top-level, beside `value`
            out["value_default_pipeline"] = body["default_pipeline"]["lm_iterations_per_s"]
        out["legs_failed"] = legs_failed
        out.update(body)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
