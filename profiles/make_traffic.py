#!/usr/bin/env python3
"""profiles/r01_pmc_summary.json -> profiles/traffic_latest.json (HBM bytes per dispatch of every kernel, gfx950-corrected:
FETCH_SIZE and WRITE_SIZE are reported in KB and FETCH_SIZE counts wide coalesced reads at half their size,
MI355X_MICROARCH.md).  usage: make_traffic.py <pmc_summary.json> > traffic_latest.json"""
import json, sys

s = json.load(open(sys.argv[1]))
out = {"_note": "rocprofv3 --pmc passes on tests/probe_run.py 256 1 (256 C2 scenes, one solve); per-dispatch means. FETCH_SIZE/WRITE_SIZE "
                "are KB; hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE reports "
                "half of wide 16-B/lane reads).",
       "kernels": {}, "slot_to_kernel": {"schur": "k_schur", "chol_syrk": "chol_update_col", "eval": "k_eval", "linearize": "k_lin_ray"}}
for k, v in s.items():
    if "FETCH_SIZE" not in v:
        continue
    out["kernels"][k] = {"fetch_bytes_raw": v["FETCH_SIZE"] * 1024, "write_bytes": v.get("WRITE_SIZE", 0.0) * 1024,
                         "hbm_bytes_corrected": (2 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024,
                         "mean_us": v.get("mean_us"), "dispatches": v.get("dispatches")}
print(json.dumps(out, indent=1))
