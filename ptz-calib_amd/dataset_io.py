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


def _online_set(rng, sc, kps, names, n_online: int, n_match: int, prefix: str = "q_"):
    """n_online query images that look from the rig centre near one of the rig's views (pan / tilt within a few degrees, other
    zoom); their matches point into extra key points APPENDED to that view's key points (kps is updated in place).
    Returns (image names, key points per image, pairs (ref image, query image, [(ref kp, query kp)]), ground-truth cameras)."""
    from . import synth
    w, h = int(sc.width), int(sc.height)
    Rlw = synth.rodrigues(sc.tlw_gt[:3]); tlw = sc.tlw_gt[3:]
    n_views = len(names)
    gt_entries = {}
    q_names, q_feats, q_pairs = [], [], []
    for q in range(n_online):
        v = int(rng.integers(0, n_views))
        cv = sc.cam_gt[v]
        Rv = synth.rodrigues(cv[4:7])
        dR = synth.rodrigues(np.deg2rad(rng.uniform(-3, 3, 3)) * np.array([1.0, 1.0, 0.2]))
        Rq = dR @ Rv
        fq = cv[0] * rng.uniform(0.8, 1.25)
        pu = rng.uniform(60, w - 60, 6 * n_match); pv = rng.uniform(60, h - 60, 6 * n_match)
        ray = np.stack([(pu - cv[2]) / cv[0], (pv - cv[3]) / cv[1], np.ones_like(pu)], 1) @ Rv  # local frame
        pc = ray @ Rq.T
        qu = fq * pc[:, 0] / pc[:, 2] + 0.5 * w; qv = fq * pc[:, 1] / pc[:, 2] + 0.5 * h
        ok = np.flatnonzero((pc[:, 2] > 0) & (qu > 10) & (qu < w - 10) & (qv > 10) & (qv < h - 10))[:n_match]
        ref_pts = np.stack([pu[ok], pv[ok]], 1) + rng.normal(size=(len(ok), 2)) * 0.5
        cur_pts = np.stack([qu[ok], qv[ok]], 1) + rng.normal(size=(len(ok), 2)) * 0.5
        off = len(kps[v])
        kps[v] = np.concatenate([kps[v], ref_pts.astype(np.float32)])
        qn = f"{prefix}{q:05d}.png"
        q_names.append(qn); q_feats.append(cur_pts.astype(np.float32))
        q_pairs.append((names[v], qn, [(off + k, k) for k in range(len(ok))]))
        cq = np.zeros(15); cq[0] = cq[1] = fq; cq[2], cq[3] = 0.5 * w, 0.5 * h
        Rw = Rq @ Rlw
        cq[4:7] = synth.rodrigues_inv(Rw); cq[7:10] = Rq @ tlw
        gt_entries[os.path.splitext(qn)[0]] = camera_json_entry(os.path.splitext(qn)[0], cq, w, h)
    return q_names, q_feats, q_pairs, gt_entries


def _write_offline(root: str, tag: str, sc, tb, kps, names) -> dict:
    """<root>/offline/<tag>/{images, <tag>.json} and <root>/offline_matches/<tag>/{features, pairs_matches.txt}; returns the
    ground-truth entries (world-frame cameras) of the rig's images."""
    from . import synth
    w, h = int(sc.width), int(sc.height)
    Rlw = synth.rodrigues(sc.tlw_gt[:3]); tlw = sc.tlw_gt[3:]
    img_dir = os.path.join(root, "offline", tag); feat_dir = os.path.join(root, "offline_matches", tag)
    os.makedirs(img_dir, exist_ok=True); os.makedirs(feat_dir, exist_ok=True)
    for n in names:
        write_png(os.path.join(img_dir, n), w, h)
    kp_ptr = np.concatenate([[0], np.cumsum([len(k) for k in kps])]).astype(np.int64)
    write_features(feat_dir, names, kp_ptr, np.concatenate(kps))
    write_matches(os.path.join(feat_dir, "pairs_matches.txt"), [(names[a], names[b], ms) for a, b, ms in tb.pairs()])
    entries, gt_entries = {}, {}
    for i, n in enumerate(names):
        sel = np.flatnonzero(sc.obs3d["cam"] == i)
        root_n = os.path.splitext(n)[0]
        cw = sc.cam_gt[i].copy()
        Ri = synth.rodrigues(cw[4:7])
        cw[4:7] = synth.rodrigues_inv(Ri @ Rlw); cw[7:10] = Ri @ tlw
        entries[root_n] = camera_json_entry(root_n, cw, w, h, sc.obs3d["uv"][sel], sc.obs3d["xyz"][sel])
        gt_entries[root_n] = camera_json_entry(root_n, cw, w, h)
    with open(os.path.join(img_dir, tag + ".json"), "w") as f:
        json.dump({"cameras": entries}, f, indent=4)
    return gt_entries


def _write_online(root: str, tag: str, sc, q_names, q_feats, q_pairs) -> None:
    on_img = os.path.join(root, "online", tag); on_feat = os.path.join(root, "online_matches", tag)
    os.makedirs(on_img, exist_ok=True); os.makedirs(on_feat, exist_ok=True)
    for n in q_names:
        write_png(os.path.join(on_img, n), int(sc.width), int(sc.height))
    q_ptr = np.concatenate([[0], np.cumsum([len(k) for k in q_feats])]).astype(np.int64)
    write_features(on_feat, q_names, q_ptr, np.concatenate(q_feats))
    write_matches(os.path.join(on_feat, "pairs_matches.txt"), q_pairs)


def write_synthetic_dataset(root: str, n_scenes: int = 10, n_views: int = 16, obs_per_view: int = 80, n_online: int = 6,
                            n_match: int = 96, seed0: int = 40) -> dict:
    """A data set in the directory layout the reference's run_ptzba_synthetic.sh / run_reloc_synthetic.sh expect:
        <root>/offline/scene_XX/{img_*.png, scene_XX.json}   images + annotation (markers on the ground plane)
        <root>/offline_matches/scene_XX/{img_*.png.txt, pairs_matches.txt}
        <root>/online/scene_XX/q_*.png, <root>/online_matches/scene_XX/{q_*.png.txt, pairs_matches.txt}
        <root>/gt/scene_XX.json                               ground-truth world-frame cameras of offline + online images
    Online images look from the rig centre near one of the offline views (pan/tilt within a few degrees, other zoom); their
    matches point into extra key points appended to that view's feature file."""
    from . import synth
    rng = np.random.default_rng(seed0)
    out = {"scenes": []}
    os.makedirs(os.path.join(root, "gt"), exist_ok=True)
    for s in range(1, n_scenes + 1):
        tag = f"scene_{s:02d}"
        sc = synth.add_annotations(synth.make_scene(seed0 + s, n_views, obs_per_view), n_annotated=8, pts_per_cam=14)
        tb = synth.make_match_table(sc, min_pair_matches=6)
        names = [image_name(i) for i in range(n_views)]
        kps = [tb.kp_xy[tb.kp_ptr[i]:tb.kp_ptr[i + 1]].astype(np.float32) for i in range(n_views)]
        q_names, q_feats, q_pairs, gt_entries = _online_set(rng, sc, kps, names, n_online, n_match)
        gt_entries.update(_write_offline(root, tag, sc, tb, kps, names))
        _write_online(root, tag, sc, q_names, q_feats, q_pairs)
        with open(os.path.join(root, "gt", tag + ".json"), "w") as f:
            json.dump({"cameras": gt_entries}, f, indent=4)
        out["scenes"].append(dict(tag=tag, n_views=n_views, n_online=n_online))
    return out


# the reference's WorldCup14 runs (run_ptzba_worldcup14.sh, run_reloc_worldcup14.sh): four recorded matches are calibrated, seven test
# sequences are relocalised against them
WORLDCUP14_MATCHES = ("GER_ARG", "GER_POR", "NED_ARG", "USA_GER")
WORLDCUP14_TESTS = (("GER_ARG", "ESP_CHI"), ("GER_ARG", "FRA_GER"), ("GER_POR", "SUI_FRA"), ("NED_ARG", "ARG_SUI"),
                    ("NED_ARG", "BRA_CRO"), ("NED_ARG", "URU_ENG"), ("USA_GER", "CRO_MEX"))


def write_worldcup14_layout(root: str, views=(20, 28, 24, 16), obs_per_view: int = 90, n_online: int = 4, n_match: int = 96,
                            seed0: int = 70) -> dict:
    """A data set in the directory layout of the reference's WorldCup14 runs -- <root>/offline/<MATCH>/, offline_matches/<MATCH>/,
    online/<TEST>/, online_matches/<TEST>/ with the reference's match and test-sequence names -- filled with synthetic rigs that
    look like a broadcast camera (1280 x 720, 120 degrees of pan, matches of different size).  The real recordings are not in
    this repository; with them under data/worldcup14 the same scripts run on them unchanged.  Also writes <root>/gt/<name>.json
    (ground-truth world-frame cameras) for the offline matches and the test sequences, which the real data set does not have in
    this form (its ground truth are homographies for the IoU evaluation)."""
    from . import synth
    rng = np.random.default_rng(seed0)
    os.makedirs(os.path.join(root, "gt"), exist_ok=True)
    rigs = {}
    for k, tag in enumerate(WORLDCUP14_MATCHES):
        sc = synth.add_annotations(synth.make_scene(seed0 + k, views[k], obs_per_view, width=1280, height=720, pan_range_deg=120.0),
                                   n_annotated=8, pts_per_cam=14)
        tb = synth.make_match_table(sc, min_pair_matches=6)
        names = [image_name(i) for i in range(views[k])]
        kps = [tb.kp_xy[tb.kp_ptr[i]:tb.kp_ptr[i + 1]].astype(np.float32) for i in range(views[k])]
        rigs[tag] = (sc, tb, names, kps)
    online = {}
    for ref, test in WORLDCUP14_TESTS:  # (every test sequence appends its reference key points to the match it is registered against)
        sc, tb, names, kps = rigs[ref]
        online[test] = (ref,) + _online_set(rng, sc, kps, names, n_online, n_match, prefix=test.lower() + "_")
    for tag, (sc, tb, names, kps) in rigs.items():
        gt = _write_offline(root, tag, sc, tb, kps, names)
        with open(os.path.join(root, "gt", tag + ".json"), "w") as f:
            json.dump({"cameras": gt}, f, indent=4)
    for test, (ref, q_names, q_feats, q_pairs, gt_entries) in online.items():
        _write_online(root, test, rigs[ref][0], q_names, q_feats, q_pairs)
        with open(os.path.join(root, "gt", test + ".json"), "w") as f:
            json.dump({"cameras": gt_entries}, f, indent=4)
    return dict(matches={t: len(rigs[t][2]) for t in rigs}, tests={t: len(online[t][1]) for t in online})
