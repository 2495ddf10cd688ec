"""Randomised parity sweep on the CPU: the device tracker programs compiled for the host (tests/_hostsim) against the numpy oracle on seeded random
scenes -- far more of them than the test suite runs.  This is how the assignment ties were found.

    python scripts/parity_sweep.py sort default 0 2000            # kind, kalman format, first seed, last seed (5..150 objects, 15..50 frames)
    python scripts/parity_sweep.py bytetrack default 0 300 --big  # 150..500 objects at 1280 px
    python scripts/parity_sweep.py deepsort default 0 700         # tests/util.py::random_deepsort_scene, oracle with the product's summation order pinned

Round 2, last run: 0 mismatches in 2000 + 2000 + 2000 + 250 small scenes (sort, bytetrack, botsort, bytetrack/strongsort), 300 + 300 + 300 + 40 big ones,
700 DeepSORT scenes.  A second pass at the end of the round on seeds nobody had looked at (sort 2000..5000, bytetrack 2000..5000, botsort 2000..4000,
deepsort 700..1600): 0 mismatches in 8900 more scenes."""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import tracker_np  # noqa: E402
from yolov7_tracker_amd import synth  # noqa: E402
from tests import _hostsim as hs  # noqa: E402
import util  # noqa: E402

kind, fmt, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
big = "--big" in sys.argv
bad, t0 = [], time.time()
for seed in range(lo, hi):
    if kind == "deepsort":
        dets, fn, dim = util.random_deepsort_scene(seed)
        want = tracker_np.run("deepsort", dets, feature_fn=fn, dot=tracker_np.dot_sequential)
        got = hs.run("deepsort", dets, feature_fn=fn, feat_dim=dim)
    else:
        rng = np.random.default_rng(seed * 7919 + zlib.crc32(kind.encode()))
        n_obj = int(rng.integers(150, 500)) if big else int(rng.integers(5, 150))
        n_frames = int(rng.integers(10, 25)) if big else int(rng.integers(15, 50))
        dets = synth.make_detections(n_frames, n_obj, 1280 if big else 640, seq_idx=1000 + seed, miss=float(rng.uniform(0, 0.3)), fp=float(rng.uniform(0, 0.25)))
        gap = int(rng.integers(0, 9))
        if gap > 2:
            dets = [None if (i % gap == gap - 1) else d for i, d in enumerate(dets)]
        warps = synth.make_warps(n_frames, seq_idx=seed) if kind == "botsort" else None
        want = tracker_np.run(kind, dets, kalman_format=fmt, warps=warps)
        got = hs.run(kind, dets, kalman_format=fmt, warps=warps)
    try:
        util.assert_same_tracks(got, want, "seed %d" % seed)
    except AssertionError as e:
        bad.append(seed)
        print("seed", seed, "MISMATCH", str(e)[:120].replace("\n", " "), flush=True)
print(kind, fmt, "big" if big else "", "seeds %d..%d: %d mismatches %s, %.0f s, assignments re-solved literally: %d" % (lo, hi, len(bad), bad, time.time() - t0,
                                                                                                         hs.lib().hs_literal_calls()))
if hs.lib().hs_next_tracker():
    print("associations on a short candidate stride %d, repeated with the full one %d" % (hs.lib().hs_next_stat(0), hs.lib().hs_next_stat(1)))
