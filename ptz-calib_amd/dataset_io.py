"""Write a synthetic rig in the on-disk layout the reference's tools read (src/core/data_io.cc): an image directory, a
feature directory with one COLMAP text file per image and pairs_matches.txt, and an annotation / camera JSON.  Used by the
tests of the run_ptz_ba / run_ptz_reloc tools; plumbing, not the product."""
from __future__ import annotations

import json
import os
import struct
import zlib

import numpy as np


def write_png(path: str, width: int, height: int) -> None:
    """A valid 8-bit grayscale PNG of the given size (all black; ~2 KB for 1920x1080)."""
    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    raw = (b"\x00" + b"\x00" * width) * height
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def image_name(i: int, ext: str = ".png") -> str:
    return f"img_{i:05d}{ext}"


def write_features(feature_dir: str, names, kp_ptr, kp_xy, desc_dim: int = 0) -> None:
    os.makedirs(feature_dir, exist_ok=True)
    for i, name in enumerate(names):
        pts = kp_xy[kp_ptr[i]:kp_ptr[i + 1]]
        with open(os.path.join(feature_dir, name + ".txt"), "w") as f:
            f.write(f"{len(pts)} {desc_dim}\n")
            for x, y in pts:
                f.write(f"{float(x):.9g} {float(y):.9g} 1 0" + " 0" * desc_dim + "\n")


def write_matches(path: str, pairs, trailing_blank: bool = True) -> None:
    """pairs: list of (name_a, name_b, [(i, j), ...]).  Without the trailing blank line the reference drops the last
    block (data_io.cc:75-86)."""
    with open(path, "w") as f:
        for k, (a, b, ms) in enumerate(pairs):
            f.write(f"{a} {b}\n")
            for i, j in ms:
                f.write(f"{int(i)} {int(j)}\n")
            if trailing_blank or k + 1 < len(pairs):
                f.write("\n")


def camera_json_entry(name_root: str, cam15, width: int, height: int, pix=(), pos=()) -> dict:
    from .synth import rodrigues
    c = np.asarray(cam15, dtype=np.float64)
    K = [c[0], 0.0, c[2], 0.0, c[1], c[3], 0.0, 0.0, 1.0]
    R = rodrigues(c[4:7])
    t = c[7:10]
    return {"name": name_root, "pos": (-R.T @ t).tolist(), "res": [int(width), int(height)], "K": K, "R": R.reshape(9).tolist(),
            "t": t.tolist(), "dist": c[10:15].tolist(), "distType": "" if c[10] < 1e-5 else "k1",
            "marker": {"pix": [[float(u) / width, float(v) / height] for u, v in pix], "pos": [list(map(float, p)) for p in pos]},
            "version": "2.0"}


def write_rig(root: str, scene, table, annotations=None, cam_for_json=None, ext: str = ".png") -> dict:
    """Layout: <root>/images/<cam_id>/img_XXXXX.png, <root>/features/*.txt + pairs_matches.txt, <root>/annotation.json.
    annotations: scene.obs3d-style dict (uv, xyz, cam) or None.  cam_for_json: cameras stored in the annotation file
    (the tools only read its markers; default: ground truth).  Returns the paths."""
    img_dir = os.path.join(root, "images", "rig0")
    feat_dir = os.path.join(root, "features")
    os.makedirs(img_dir, exist_ok=True)
    names = [image_name(i, ext) for i in range(table.n_img)]
    for i, n in enumerate(names):
        write_png(os.path.join(img_dir, n), int(table.img_wh[i, 0]), int(table.img_wh[i, 1]))
    write_features(feat_dir, names, table.kp_ptr, table.kp_xy)
    pairs = [(names[s], names[d], ms) for s, d, ms in table.pairs()]
    write_matches(os.path.join(feat_dir, "pairs_matches.txt"), pairs)
    cams = scene.cam_gt if cam_for_json is None else cam_for_json
    entries = {}
    for i, n in enumerate(names):
        pix, pos = [], []
        if annotations is not None:
            sel = np.flatnonzero(annotations["cam"] == i)
            pix = annotations["uv"][sel]; pos = annotations["xyz"][sel]
        entries[os.path.splitext(n)[0]] = camera_json_entry(os.path.splitext(n)[0], cams[i], table.img_wh[i, 0], table.img_wh[i, 1], pix, pos)
    annot = os.path.join(root, "annotation.json")
    with open(annot, "w") as f:
        json.dump({"cameras": entries}, f, indent=4)
    return dict(images=img_dir, features=feat_dir, annotation=annot, names=names)


def write_reloc_set(root: str, rb, width: int = 1920, height: int = 1080) -> dict:
    """A relocalization data set for run_ptz_reloc from a synth.RelocBatch: reference image q / test image q per query,
    <root>/ref_images/refs, <root>/ref_features, <root>/ref_params.json, <root>/test_images/tests, <root>/test_features
    (with pairs_matches.txt: blocks 'ref_name test_name')."""
    ref_img = os.path.join(root, "ref_images", "refs"); test_img = os.path.join(root, "test_images", "tests")
    ref_feat = os.path.join(root, "ref_features"); test_feat = os.path.join(root, "test_features")
    for d in (ref_img, test_img, ref_feat, test_feat):
        os.makedirs(d, exist_ok=True)
    ref_names = [f"ref_{q:05d}.png" for q in range(rb.n_query)]
    test_names = [f"test_{q:05d}.png" for q in range(rb.n_query)]
    for n in ref_names:
        write_png(os.path.join(ref_img, n), width, height)
    for n in test_names:
        write_png(os.path.join(test_img, n), width, height)
    write_features(ref_feat, ref_names, rb.match_ptr, rb.uv_ref)
    write_features(test_feat, test_names, rb.match_ptr, rb.uv_cur)
    pairs = [(ref_names[q], test_names[q], [(k, k) for k in range(int(rb.match_ptr[q + 1] - rb.match_ptr[q]))]) for q in range(rb.n_query)]
    write_matches(os.path.join(test_feat, "pairs_matches.txt"), pairs)
    entries = {os.path.splitext(n)[0]: camera_json_entry(os.path.splitext(n)[0], rb.cam_ref[q], width, height) for q, n in enumerate(ref_names)}
    params = os.path.join(root, "ref_params.json")
    with open(params, "w") as f:
        json.dump({"cameras": entries}, f, indent=4)
    return dict(ref_images=ref_img, ref_features=ref_feat, ref_params=params, test_images=test_img, test_features=test_feat,
                ref_names=ref_names, test_names=test_names)
