#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch and
mean kernel duration (from the kernel trace of the same run).  usage: summarize_pmc.py <dir> [<dir> ...]"""
import csv, glob, os, sys, collections, json

def short(name):
    for key in ("k_schur", "k_lin_ray", "k_lin_cam", "k_eval", "k_ray_prep", "k_cam_prep", "k_cam_update", "k_cam_diag", "k_lm_pre", "k_lm_post",
                "chol_update_col", "chol_col_step", "chol_tile_inverse", "chol_clear_tiles", "chol_syrk", "chol_trsm", "chol_diag", "chol_backsolve", "chol_pad", "k_krt", "k_reset",
                "k_fill", "k_jacobi", "k_compact", "k_ctl_reset", "fillBuffer", "pat_stream16", "pat_stream8", "pat_stream4", "pat_rows96", "pat_gather96", "pat_gather64"):
        if key in name:
            return key
    return name[:40]

def main():
    out = {}
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
            for k, cs in acc.items():
                for c, (tot, n) in cs.items():
                    out.setdefault(k, {})[c] = tot / n
                    out[k]["dispatches"] = n
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            dur = collections.defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])); dur[k][1] += 1
            for k, (t, n) in dur.items():
                out.setdefault(k, {})["mean_us"] = t / n / 1e3
    print(json.dumps(out, indent=1, sort_keys=True))

if __name__ == "__main__":
    main()
