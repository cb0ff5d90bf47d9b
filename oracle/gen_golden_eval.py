#!/usr/bin/env python3
"""Generates tests/golden/eval_synthetic_vectors.json by IMPORTING the reference's scripts/eval_synthetic.py
(from /root/reference, in the build container only) and recording inputs/outputs of its pure functions
calc_focal_error (:36-38), calc_ape (:41-65) and cal_mean_median (:68-74).  The fixture is data only."""
import importlib.util
import json
import math
import os
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scripts/eval_synthetic.py"


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K


def main():
    spec = importlib.util.spec_from_file_location("ref_eval_synthetic", REF)
    ref = importlib.util.module_from_spec(spec)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spec.loader.exec_module(ref)
    rng = np.random.default_rng(20250217)
    cases = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(24):
            Rp = rot(rng.standard_normal(3), rng.uniform(0, 3.1))
            Rg = Rp @ rot(rng.standard_normal(3), rng.uniform(0, 0.2 if k % 3 else 3.0))
            tp, tg = rng.standard_normal(3) * 5, rng.standard_normal(3) * 5
            if k == 0:
                Rg, tg = Rp.copy(), tp.copy()  # identical poses
            if k == 1:
                tp, tg = np.zeros(3), np.zeros(3)  # rotation-only rig (the PTZ case)
            fp, fg = rng.uniform(1500, 4000), rng.uniform(1500, 4000)
            at, ar = ref.calc_ape(Rp, tp.reshape(3, 1), Rg, tg.reshape(3, 1))
            cases.append(dict(pred_R=Rp.tolist(), pred_t=tp.tolist(), gt_R=Rg.tolist(), gt_t=tg.tolist(), pred_f=fp, gt_f=fg,
                              focal_error=float(ref.calc_focal_error(fp, fg)), ape_trans=float(at), ape_rot_deg=float(ar)))
        lists = [[1.0, 2.0, 4.0], [3.5], [1.0, float("nan"), 5.0, 2.0], [0.25, 0.75, float("nan"), float("nan")]]
        mm = []
        for lst in lists:
            m, md = ref.cal_mean_median(lst)
            mm.append(dict(data=[None if (isinstance(v, float) and math.isnan(v)) else v for v in lst], mean=float(m), median=float(md)))
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "eval_synthetic_vectors.json")
    with open(out, "w") as f:
        json.dump({"_generator": "oracle/gen_golden_eval.py importing /root/reference/scripts/eval_synthetic.py",
                   "ape_cases": cases, "mean_median_cases": mm}, f, separators=(",", ":"))
    print("wrote", out, len(cases), "pose cases")


if __name__ == "__main__":
    main()
