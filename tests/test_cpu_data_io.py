"""CPU tests of the on-disk formats (ptz-calib_amd/host/data_io.cc, json_mini.cc, image_size.cc, homography.cc): the
reference's readers/writers (src/core/data_io.cc) restated without OpenCV / nlohmann."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

import host_util as hu


def probe(cmd, a="", b=""):
    lib = hu.lib()
    lib.ptzh_io_probe.restype = C.c_void_p
    p = lib.ptzh_io_probe(cmd.encode(), a.encode(), b.encode())
    txt = C.string_at(p).decode()
    lib.ptzh_free(C.c_void_p(p))
    return json.loads(txt)


def test_json_parse_and_dump_layout():
    """Writer lays text out like nlohmann's dump(4): insertion order, 4 spaces, shortest round-trip numbers, '.0' on
    integral floats, integers stay integers."""
    src = '{"b": [1, 2.0, 2304.0, 0.1, -1e-05, 1e+20, 1.5e-300], "a": {"s": "x\\"y\\n\\u00e9", "t": true, "n": null, "e": [], "o": {}}}'
    r = probe("json", src)
    assert r["ok"]
    assert json.loads(r["dump"]) == json.loads(src)
    assert r["dump"].index('"b"') < r["dump"].index('"a"')  # insertion order kept
    assert '\n    "b": [\n        1,\n        2.0,\n        2304.0,\n        0.1,\n        -1e-05,\n        1e+20,\n        1.5e-300\n    ]' in r["dump"]
    assert '"e": [],' in r["dump"] and '"o": {}' in r["dump"]
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.normal(size=50) * 10.0 ** rng.integers(-8, 8, 50), np.float32(rng.normal(size=20)).astype(np.float64)])
    r = probe("json", json.dumps({"v": vals.tolist()}))
    assert np.array_equal(np.array(json.loads(r["dump"])["v"]), vals)  # exact round trip of every double
    for bad in ['{"a": 1,}', '[1 2]', '{"a" 1}', '"unterminated', '{"a": tru}', '']:
        assert not probe("json", bad)["ok"]


def _png(path, w, h):
    from importlib import import_module
    import __graft_entry__ as ge
    ge.load_package().dataset_io.write_png(path, w, h)


def test_image_size_from_headers(tmp_path):
    """cv::imread(path).size() replacement (data_io.cc:316-322): PNG IHDR, JPEG SOF (after APP segments), BMP, TIFF."""
    p = str(tmp_path / "a.png"); _png(p, 1920, 1080)
    assert probe("image_size", p) == {"ok": True, "width": 1920, "height": 1080}
    # JPEG: SOI, APP0 (16 bytes), DQT stub, SOF0 with 720 x 1280, then garbage
    jpg = b"\xff\xd8" + b"\xff\xe0" + struct.pack(">H", 16) + b"JFIF\x00" + b"\x00" * 9 + b"\xff\xdb" + struct.pack(">H", 4) + b"\x00\x00" + \
          b"\xff\xc0" + struct.pack(">HBHHB", 11, 8, 720, 1280, 1) + b"\x01\x11\x00" + b"\xff\xd9"
    p = str(tmp_path / "b.jpg"); open(p, "wb").write(jpg)
    assert probe("image_size", p) == {"ok": True, "width": 1280, "height": 720}
    bmp = b"BM" + struct.pack("<IHHI", 54, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 640, -480, 1, 24, 0, 0, 0, 0, 0, 0)
    p = str(tmp_path / "c.bmp"); open(p, "wb").write(bmp)
    assert probe("image_size", p) == {"ok": True, "width": 640, "height": 480}
    tif = b"II" + struct.pack("<HI", 42, 8) + struct.pack("<H", 2) + struct.pack("<HHII", 256, 3, 1, 800) + struct.pack("<HHII", 257, 4, 1, 600) + b"\x00" * 4
    p = str(tmp_path / "d.tiff"); open(p, "wb").write(tif)
    assert probe("image_size", p) == {"ok": True, "width": 800, "height": 600}
    p = str(tmp_path / "e.png"); open(p, "wb").write(b"not an image")
    assert not probe("image_size", p)["ok"] and not probe("image_size", str(tmp_path / "missing.png"))["ok"]


def test_load_images_features_and_match_table(pkg, tmp_path):
    """LoadImgsAndFeatures + LoadMatchesInfo (data_io.cc:294-400): sorted listing, extension filter, mask.png skipped, unreadable
    images skipped, N x N table with the listed direction only, confidence = min(1, n / 100) in float, and the quirk that a final
    block without a trailing blank line is dropped (:75-86)."""
    dio = pkg.dataset_io
    sc = pkg.synth.make_scene(3, 8, 60)
    tb = pkg.synth.make_match_table(sc, bidirectional=False, min_pair_matches=4)
    paths = dio.write_rig(str(tmp_path), sc, tb)
    img = paths["images"]
    dio.write_png(os.path.join(img, "mask.png"), 10, 10)            # skipped by name
    open(os.path.join(img, "notes.txt"), "w").write("x")             # skipped by extension
    open(os.path.join(img, "zzz_broken.png"), "w").write("garbage")  # unreadable -> skipped
    r = probe("load", img, paths["features"])
    assert r["ok"] and r["fnames"] == paths["names"]
    assert r["sizes"] == [[1920, 1080]] * 8
    assert r["n_keypoints"] == np.diff(tb.kp_ptr).tolist()
    assert np.allclose(r["first_keypoint"][2], tb.kp_xy[tb.kp_ptr[2]], rtol=0, atol=1e-3)
    assert r["table_cells"] == 64
    got = {(p["src"], p["dst"]): p for p in r["pairs"]}
    assert sorted(got) == sorted(zip(tb.src.tolist(), tb.dst.tolist()))
    for k, (s, d) in enumerate(zip(tb.src.tolist(), tb.dst.tolist())):
        p = got[(s, d)]
        n = int(tb.match_ptr[k + 1] - tb.match_ptr[k])
        assert p["n"] == n and not p["H_empty"]
        assert p["confidence"] == float(np.float32(1.0) if n >= 100 else np.float32(n) / np.float32(100))
        assert p["first_match"] == [int(tb.q[tb.match_ptr[k]]), int(tb.t[tb.match_ptr[k]])]
        H = np.array(p["H"]).reshape(3, 3)
        assert abs(H[2, 2] - 1) < 1e-12
        # outlier-free matches: RANSAC + refit lands on the least-squares homography of synth.homography_dlt
        a = tb.kp_xy[tb.kp_ptr[s] + tb.q[tb.match_ptr[k]:tb.match_ptr[k + 1]]].astype(np.float64)
        b = tb.kp_xy[tb.kp_ptr[d] + tb.t[tb.match_ptr[k]:tb.match_ptr[k + 1]]].astype(np.float64)
        ph = np.c_[a, np.ones(len(a))] @ H.T
        err = np.sqrt((((ph[:, :2] / ph[:, 2:]) - b) ** 2).sum(1))
        assert err.max() < 4.0 and np.sqrt((err ** 2).mean()) < 1.6  # 0.5 px noise per coordinate on both images
    # the same file without the closing blank line loses its last block
    pairs = [(paths["names"][s], paths["names"][d], ms) for s, d, ms in tb.pairs()]
    dio.write_matches(os.path.join(paths["features"], "pairs_matches.txt"), pairs, trailing_blank=False)
    r2 = probe("load", img, paths["features"])
    assert len(r2["pairs"]) == len(r["pairs"]) - 1
    assert (int(tb.src[-1]), int(tb.dst[-1])) not in {(p["src"], p["dst"]) for p in r2["pairs"]}


def test_camera_json_roundtrip_and_annotation(pkg, tmp_path):
    """ReadFromJson / SaveToJson / LoadAnnotation (data_io.cc:112-295, 403-432): pixel markers are stored normalised by the image
    size, names are matched without extension, objects are read in key order, written in insertion order."""
    dio = pkg.dataset_io
    sc = pkg.synth.add_annotations(pkg.synth.make_scene(2, 8, 60))
    tb = pkg.synth.make_match_table(sc, min_pair_matches=4)
    paths = dio.write_rig(str(tmp_path), sc, tb, annotations=sc.obs3d)
    r = probe("annotation", paths["annotation"], paths["images"])
    assert r["ok"] and len(r["pixels"]) == 8
    for i in range(8):
        sel = np.flatnonzero(sc.obs3d["cam"] == i)
        assert len(r["pixels"][i]) == len(sel)
        if len(sel):
            assert np.abs(np.array(r["pixels"][i]) - sc.obs3d["uv"][sel]).max() < 2e-3  # float32 pixel = width * (u / width)
            assert np.array_equal(np.array(r["pts3d"][i]), sc.obs3d["xyz"][sel])
    out = str(tmp_path / "rewritten.json")
    rr = probe("rewrite", paths["annotation"], out)
    assert rr["ok"] and rr["names"] == sorted(os.path.splitext(n)[0] for n in paths["names"])
    a = json.load(open(paths["annotation"]))["cameras"]; b = json.load(open(out))["cameras"]
    assert list(b.keys()) == rr["names"]
    for name in a:
        assert list(b[name].keys()) == ["name", "pos", "res", "K", "R", "t", "dist", "distType", "marker", "version"]
        for key in ("K", "R", "t", "dist", "res"):
            assert b[name][key] == a[name][key]
        assert np.allclose(b[name]["pos"], a[name]["pos"], atol=1e-9)
        assert np.allclose(np.array(b[name]["marker"]["pix"]).reshape(-1, 2), np.array(a[name]["marker"]["pix"]).reshape(-1, 2), atol=1e-6)
        assert b[name]["marker"]["pos"] == a[name]["marker"]["pos"] and b[name]["version"] == "2.0"
    assert not probe("rewrite", str(tmp_path / "missing.json"), out)["ok"]
    open(str(tmp_path / "bad.json"), "w").write('{"cameras": {"x": {"K": [1, 2]}}}')
    assert not probe("rewrite", str(tmp_path / "bad.json"), out)["ok"]


def test_ransac_homography_rejects_outliers():
    """findHomography(RANSAC, 4 px) replacement: 30 % gross outliers are rejected, the inlier fit recovers H; fewer than four
    correspondences give the empty matrix."""
    rng = np.random.default_rng(3)
    H = np.array([[1.02, 0.03, 40.0], [-0.02, 0.98, -25.0], [1e-5, -2e-5, 1.0]])
    n = 200
    a = np.c_[rng.uniform(0, 1920, n), rng.uniform(0, 1080, n)]
    ph = np.c_[a, np.ones(n)] @ H.T
    b = ph[:, :2] / ph[:, 2:] + rng.normal(size=(n, 2)) * 0.5
    out_idx = rng.choice(n, 60, replace=False)
    b[out_idx] += rng.uniform(30, 300, (60, 2)) * rng.choice([-1, 1], (60, 2))
    a32 = np.ascontiguousarray(a, dtype=np.float32); b32 = np.ascontiguousarray(b, dtype=np.float32)
    Hout = np.zeros(9); mask = np.zeros(n, dtype=np.uint8)
    lib = hu.lib()
    ok = lib.ptzh_find_homography(n, hu._p(a32), hu._p(b32), C.c_double(4.0), hu._p(Hout), hu._p(mask))
    assert ok == 1
    inl = np.setdiff1d(np.arange(n), out_idx)
    assert mask[out_idx].sum() == 0 and mask[inl].mean() > 0.97
    Hn = Hout.reshape(3, 3)
    corners = np.array([[0, 0, 1], [1920, 0, 1], [0, 1080, 1], [1920, 1080, 1.0]])
    p1 = corners @ Hn.T; p2 = corners @ H.T
    assert np.abs(p1[:, :2] / p1[:, 2:] - p2[:, :2] / p2[:, 2:]).max() < 0.5
    assert lib.ptzh_find_homography(3, hu._p(a32), hu._p(b32), C.c_double(4.0), hu._p(Hout), None) == 0
