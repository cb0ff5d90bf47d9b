"""Accuracy yardstick of the synthetic benchmarks: the metrics of the reference's scripts/eval_synthetic.py
(focal absolute error :36-38, absolute pose error :41-65, nan-aware mean/median :68-74), written independently
and pinned against vectors recorded from the reference script (tests/golden/eval_synthetic_vectors.json)."""
from __future__ import annotations

import math

import numpy as np


def calc_focal_error(pred_f: float, gt_f: float) -> float:
    return abs(pred_f - gt_f)


def calc_ape(pred_R, pred_t, gt_R, gt_t):
    """APE of P_pred P_gt^-1: translation |-(R_rel)^T t_rel| and rotation angle in degrees."""
    pred_R, gt_R = np.asarray(pred_R, float).reshape(3, 3), np.asarray(gt_R, float).reshape(3, 3)
    pred_t, gt_t = np.asarray(pred_t, float).reshape(3), np.asarray(gt_t, float).reshape(3)
    R_rel = pred_R @ gt_R.T                 # [R_p t_p][R_g t_g]^-1 = [R_p R_g^T, t_p - R_p R_g^T t_g]
    t_rel = pred_t - R_rel @ gt_t
    ape_trans = float(np.linalg.norm(-R_rel.T @ t_rel))
    # rotation-vector norm of R_rel (scipy Rotation.as_rotvec): robust angle from the skew part and the trace
    s = 0.5 * np.linalg.norm([R_rel[2, 1] - R_rel[1, 2], R_rel[0, 2] - R_rel[2, 0], R_rel[1, 0] - R_rel[0, 1]])
    c = 0.5 * (np.trace(R_rel) - 1.0)
    return ape_trans, math.degrees(math.atan2(s, c))


def cal_mean_median(data):
    a = np.array([np.nan if v is None else v for v in data], dtype=float)
    return float(np.nanmean(a)), float(np.nanmedian(a))


def evaluate_cameras(cam_pred: np.ndarray, cam_gt: np.ndarray, rodrigues):
    """Per-camera focal error and APE for 15-vector cameras; `rodrigues` maps rvec -> 3x3.  Rotation-only rigs are
    compared after removing the global gauge rotation (R_pred,0^T R_gt,0), as the solved frame is arbitrary."""
    Rp = [rodrigues(c[4:7]) for c in cam_pred]
    Rg = [rodrigues(c[4:7]) for c in cam_gt]
    G = Rp[0].T @ Rg[0]
    fe, ar, at = [], [], []
    for i in range(len(cam_pred)):
        fe.append(calc_focal_error(cam_pred[i, 0], cam_gt[i, 0]))
        t, r = calc_ape(Rp[i] @ G, cam_pred[i, 7:10], Rg[i], cam_gt[i, 7:10])
        at.append(t)
        ar.append(r)
    return dict(focal_error_abs=cal_mean_median(fe), ape_rot_deg=cal_mean_median(ar), ape_trans=cal_mean_median(at))
