#!/usr/bin/env python3
"""PMC summaries -> profiles/traffic_latest.json: HBM bytes per dispatch of every kernel family of bench.py's profile.

FETCH_SIZE / WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE counts a STREAMED read at half its size and a GATHERED one at
its size -- calibrated on this pool's MI355X with tools/probes/hip/fetch_calib.hip (profiles/r02_fetch_calibration.json: 0.5000 for
16-, 8- and 4-byte-per-lane unit-stride streams and for one 96-byte row per lane in row order; 1.00 / 1.02 for 96-byte rows /
64-byte records at random positions).  So the x2 of MI355X_MICROARCH.md applies to the streamed part of a kernel's reads only:
    true_fetch = raw + min(raw, streamed / 2) ... with `streamed` = the bytes the kernel is KNOWN to read as streams,
from the data layout (below, per active scene and LM pass) x the mean number of active scenes per dispatch of the profiled
solve.  Kernels without a stream model get the bounds [raw, 2 raw] and the midpoint as the estimate.
usage: make_traffic.py <pmc_summary.json> [...] --lm-steps <LM steps of the profiled solve> --tag C4:1000x200x500 > traffic_latest.json"""
import argparse, json

ap = argparse.ArgumentParser()
ap.add_argument("summaries", nargs="+")
ap.add_argument("--lm-steps", type=float, required=True)
ap.add_argument("--jac-evals", type=float, default=None)
ap.add_argument("--tag", required=True)
ap.add_argument("--n-obs", type=float, default=100440.0)
ap.add_argument("--n-ray", type=float, default=13432.0)
ap.add_argument("--n-ent", type=float, default=470000.0)
ap.add_argument("--schur-w", action="store_true", help="the profiled library ran with PTZ_BA_SCHUR_W=1 (round 2's k_schur_w)")
args = ap.parse_args()
s = {}
for f in args.summaries:
    for k, v in json.load(open(f)).items():
        s.setdefault(k, {}).update(v)

# bytes a kernel reads as STREAMS per active scene and pass (C2-shaped scene; data layout of DESIGN.md section 3)
STREAMS = {
    # round 3: no W rows any more -- ray ids (4 B) per observation, 4-byte entry records, 8-byte run records (<= 256 per view);
    # the per-ray records (round 5: six 16-byte pieces of planes) are gathered.  (--schur-w: round 2's kernel, a camera's W rows (96 B) streamed as well)
    "k_schur": (100.0 if args.schur_w else 4.0) * args.n_obs + 4.0 * args.n_ent + (0.0 if args.schur_w else 8.0 * 256 * 200),
    "k_lin_cam": 12.0 * args.n_obs,                               # pixel (8 B) + ray id (4 B) per observation; the 64-byte ray records are gathered
    "k_eval": 2 * 16.0 * args.n_obs + 248.0 * args.n_ray,         # observation records twice, per-ray arrays (round 5: E is three unit-stride planes, 48 B)
    "k_lin_ray": 16.0 * args.n_obs + 100.0 * args.n_ray,
    "k_ray_prep": 160.0 * args.n_ray,                             # V (48), g_r (24), the ray's record of the linearisation (64), the LM diagonal (24)
}
FAMILY = {"schur": ["k_schur"], "linearize": ["k_lin_ray", "k_lin_cam"], "eval": ["k_eval"], "ray_prep": ["k_ray_prep"],
          "chol_syrk": ["chol_update_col"], "chol_panel": ["chol_trsm"], "chol_backsolve": ["chol_backsolve", "chol_tile_inverse"]}
out = {"workload_tag": args.tag,
       "_note": "rocprofv3 --pmc passes on tools/probes/probe_c4pmc.py (bench.py's C4 workload, one solve, one scene group); per-dispatch means over every "
                "dispatch of the solve, thin passes included.  hbm_bytes = calibrated fetch + WRITE_SIZE; see make_traffic.py for the calibration.",
       "kernels": {}, "families": {}}
for k, v in s.items():
    if "FETCH_SIZE" not in v:
        continue
    raw = v["FETCH_SIZE"] * 1024.0
    wr = v.get("WRITE_SIZE", 0.0) * 1024.0
    n = v.get("dispatches", 1)
    row = {"fetch_bytes_raw": raw, "write_bytes": wr, "dispatches": n, "mean_us": v.get("mean_us"),
           "hbm_bytes_lo": raw + wr, "hbm_bytes_hi": 2 * raw + wr}
    if k in STREAMS:
        units = (args.jac_evals if (k.startswith("k_lin") and args.jac_evals) else args.lm_steps) / n  # active scenes per dispatch
        streamed = STREAMS[k] * units
        row["streamed_bytes_model"] = streamed
        row["hbm_bytes"] = raw + min(raw, streamed / 2.0) + wr
        row["note"] = "calibrated: streamed part counted at 1/2 by FETCH_SIZE, gathered part at 1"
    else:
        row["hbm_bytes"] = 1.5 * raw + wr
        row["note"] = "midpoint of [raw, 2 x raw] fetch + writes (no stream model for this kernel; tile copies are 16-byte streams -> nearer the upper bound)"
        if k.startswith("chol_"):
            row["hbm_bytes"] = 2 * raw + wr
            row["note"] = "16-byte-per-lane tile streams: FETCH_SIZE x 2 (calibrated) + writes"
    if "TCC_HIT_sum" in v:
        row["l2_hit_rate"] = v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0.0), 1.0)
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if c in v and v.get("SQ_WAVE_CYCLES"):
            row[c.lower() + "_frac"] = v[c] / v["SQ_WAVE_CYCLES"]
    if v.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("mean_us"):
        # busy cycles summed over the chip's 1024 SIMDs; dispatch time x 2.4 GHz x 1024 = the cycles available
        row["mfma_busy_frac_of_chip"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["mean_us"] * 1e-6 * 2.4e9 * 1024)
    out["kernels"][k] = row
for fam, ks in FAMILY.items():
    rows = [out["kernels"][k] for k in ks if k in out["kernels"]]
    if not rows:
        continue
    # bytes per FAMILY launch group: the kernels of a family run once per pass each (chol kernels: per block column)
    tot = sum(r["hbm_bytes"] * r["dispatches"] for r in rows)
    disp = max(r["dispatches"] for r in rows)
    out["families"][fam] = {"hbm_bytes": tot / sum(r["dispatches"] for r in rows) if fam.startswith("chol") else tot / disp,
                            "note": "; ".join(f"{k}: {out['kernels'][k]['note']}" for k in ks if k in out["kernels"])}
print(json.dumps(out, indent=1))
