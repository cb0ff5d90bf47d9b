#!/usr/bin/env python3
"""Generates tests/golden/tracks_*.json from oracle/_ref/ref_tracks, i.e. from the REFERENCE's own
union_find.h + flat_pair_map.h (compiled from /root/reference by oracle/Makefile).  Runs in the build
container only (the reference tree is not present on the GPU box); the fixtures are data: match lists in,
{track_id: {image: feature}} out.

Cases cover what tracks.cc handles: chains, union-by-rank tie breaks, path compression order, a track with
a repeated image (rejected, tracks.cc:77), tracks shorter than Filter(4) (ptzray_optimizer.cc:541), empty
match lists, self-consistent cycles, and seeded random match graphs.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_py as orc  # noqa: E402


class Lcg:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFF

    def next(self, n):
        self.s = (1664525 * self.s + 1013904223) & 0xFFFFFFFF
        return (self.s >> 8) % n


def random_case(seed, n_img, n_feat, n_pairs, max_matches):
    r = Lcg(seed)
    pairs = []
    for _ in range(n_pairs):
        i = r.next(n_img)
        j = r.next(n_img)
        if i == j:
            j = (j + 1) % n_img
        m = []
        for _ in range(r.next(max_matches + 1)):
            m.append((r.next(n_feat), r.next(n_feat)))
        pairs.append((i, j, m))
    return pairs


def ring_case(n_img, n_tracks, span):
    """Consistent multi-view tracks on a ring of images: track t is seen by `span` consecutive images."""
    pairs = {}
    for t in range(n_tracks):
        start = (7 * t) % n_img
        imgs = [(start + k) % n_img for k in range(span)]
        for a in range(len(imgs)):
            for b in range(a + 1, len(imgs)):
                i, j = imgs[a], imgs[b]
                pairs.setdefault((i, j), []).append((t, t))
    return [(i, j, m) for (i, j), m in sorted(pairs.items())]


CASES = {
    "chain4": ([(0, 1, [(0, 0)]), (1, 2, [(0, 3)]), (2, 3, [(3, 5)])], 4),
    "too_short": ([(0, 1, [(0, 0)]), (1, 2, [(0, 3)])], 4),
    "min2": ([(0, 1, [(0, 0)]), (1, 2, [(0, 3)]), (4, 5, [(1, 1)])], 2),
    "repeated_image": ([(0, 1, [(0, 0), (2, 0)]), (1, 2, [(0, 3)]), (2, 3, [(3, 5)]), (3, 4, [(5, 1)])], 4),
    "rank_ties": ([(0, 1, [(0, 0)]), (2, 3, [(0, 0)]), (1, 2, [(0, 0)]), (4, 5, [(0, 0)]), (6, 7, [(0, 0)]), (5, 6, [(0, 0)]),
                   (3, 4, [(0, 0)])], 4),
    "empty_lists": ([(0, 1, []), (1, 2, [(4, 4)]), (2, 3, [(4, 4)]), (3, 0, [(4, 9)]), (0, 2, [])], 3),
    "ring_24x40_span5": (ring_case(24, 40, 5), 4),
    "random_a": (random_case(11, 12, 30, 40, 25), 4),
    "random_b": (random_case(12, 30, 12, 120, 10), 4),
    "random_c_min3": (random_case(13, 8, 200, 28, 60), 3),
}


def main():
    assert orc.ref_tracks_available(), "run `make -C oracle ref` in the build container first"
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    doc = {"_generator": "oracle/gen_golden_tracks.py via oracle/_ref/ref_tracks "
                         "(reference src/core/union_find.h + flat_pair_map.h, Build/Filter/ExportToSTL of tracks.cc:19-113)",
           "cases": {}}
    for name, (pairs, min_len) in CASES.items():
        tracks = orc.ref_tracks_build(pairs, min_len)
        doc["cases"][name] = {
            "min_track_length": min_len,
            "pairs": [[i, j, [list(m) for m in ms]] for i, j, ms in pairs],
            "tracks": {str(t): {str(img): feat for img, feat in sorted(v.items())} for t, v in sorted(tracks.items())},
        }
        print(name, "->", len(tracks), "tracks")
    with open(os.path.join(out_dir, "tracks_reference.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
