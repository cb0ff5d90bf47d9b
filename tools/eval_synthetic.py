#!/usr/bin/env python3
"""Accuracy report of a result file against ground truth, with the metrics and the output lines of the reference's
scripts/eval_synthetic.py (focal error, absolute pose error of P_pred P_gt^-1 in degrees / metres, nan-aware mean and median).
The metric functions live in ptz-calib_amd/evalmetrics.py and are pinned against vectors generated from the reference script
(tests/golden/eval_synthetic_vectors.json).   usage: tools/eval_synthetic.py --pred out.json --gt gt.json"""
import argparse, json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="Evaluate results on synthetic dataset")
    ap.add_argument("--pred", required=True, help="Path to prediction file")
    ap.add_argument("--gt", required=True, help="Path to ground-truth file")
    args = ap.parse_args()
    em = ge.load_package().evalmetrics
    pred = json.load(open(args.pred, encoding="utf-8"))["cameras"]
    gt = json.load(open(args.gt, encoding="utf-8"))["cameras"]
    fe, ar, at = [], [], []
    for key in pred:
        K = np.array(pred[key]["K"]).reshape(3, 3); R = np.array(pred[key]["R"]).reshape(3, 3); t = np.array(pred[key]["t"]).reshape(3, 1)
        Kg = np.array(gt[key]["K"]).reshape(3, 3); Rg = np.array(gt[key]["R"]).reshape(3, 3); tg = np.array(gt[key]["t"]).reshape(3, 1)
        fe.append(em.calc_focal_error(K[0, 0], Kg[0, 0]))
        trans, rot = em.calc_ape(R, t, Rg, tg)
        ar.append(rot); at.append(trans)
    print(f"Total sample number: {len(pred)}")
    print(f"focal_error_abs [mean, median]: {em.cal_mean_median(fe)[0] :.2f}, {em.cal_mean_median(fe)[1] :.2f}")
    print(f"ape_rot [mean, median]: {em.cal_mean_median(ar)[0]: .2f}, {em.cal_mean_median(ar)[1]: .2f}")
    print(f"ape_trans [mean, median]: {em.cal_mean_median(at)[0]: .2f}, {em.cal_mean_median(at)[1]: .2f}")


if __name__ == "__main__":
    main()
