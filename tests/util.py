"""shared helpers of the test-suite (golden fixture loading, track-stream comparison)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRACKER_CASES = ["sort_default", "bytetrack_default", "bytetrack_default_gaps", "bytetrack_botsort", "sort_strongsort",
                 "bytetrack_crowd", "botsort_gmc", "botsort_crowd"]
DEEPSORT_CASES = ["deepsort_default", "deepsort_crowd", "deepsort_dim512", "deepsort_dim100"]      # appearance features from synth.make_features at the get_feature seam
ORACLE_ONLY_CASES = DEEPSORT_CASES
# stated tolerance: ids / cls identical, tlwh within 1e-6 relative (scale: image size ~1e3 px)
TLWH_RTOL, TLWH_ATOL = 1e-6, 1e-5


def load_tracker_case(name):
    g = np.load(os.path.join(GOLDEN, "tracker_%s.npz" % name))
    counts = g["det_counts"]
    dets, off = [], 0
    for c in counts:
        if c < 0:
            dets.append(None)
        else:
            dets.append(g["dets"][off:off + c])
            off += c
    frames = [[] for _ in counts]
    for f, i, b, c, s in zip(g["frame"], g["track_id"], g["tlwh"], g["cls"], g["score"]):
        frames[f].append((int(i), b, float(c), float(s)))
    return str(g["tracker"]), str(g["kalman_format"]), dets, frames


def tracker_feat_dim(name):
    """embedding width a DeepSORT case was recorded with (synth.make_features(boxes, dim=...))"""
    g = np.load(os.path.join(GOLDEN, "tracker_%s.npz" % name))
    return int(g["feat_dim"]) if "feat_dim" in g.files else 128


def feature_fn_for(name):
    from yolov7_tracker_amd import synth
    dim = tracker_feat_dim(name)
    return lambda boxes: synth.make_features(boxes, dim=dim)


def load_tracker_warps(name):
    """(n_frames, 2, 3) camera-motion matrices of a BoT-SORT case, or None"""
    g = np.load(os.path.join(GOLDEN, "tracker_%s.npz" % name))
    w = g["warps"] if "warps" in g.files else np.zeros((0, 2, 3))
    return w if len(w) else None


def assert_same_tracks(got, want, what=""):
    assert len(got) == len(want), what
    for f, (a, b) in enumerate(zip(got, want)):
        ia, ib = [r[0] for r in a], [r[0] for r in b]
        assert ia == ib, "%s frame %d: ids differ\n got  %s\n want %s" % (what, f, ia[:20], ib[:20])
        for ra, rb in zip(a, b):
            np.testing.assert_allclose(np.asarray(ra[1], dtype=np.float64), rb[1], rtol=TLWH_RTOL, atol=TLWH_ATOL,
                                       err_msg="%s frame %d id %d tlwh" % (what, f, ra[0]))
            assert float(ra[2]) == float(rb[2]), "%s frame %d id %d cls" % (what, f, ra[0])
            assert abs(float(ra[3]) - float(rb[3])) < 1e-6, "%s frame %d id %d score" % (what, f, ra[0])


def unique_optimum(cost, limit, x):
    """True when the optimal partial matching of lap's extended problem is unique (checked by forbidding
    each kept pair / each unmatched decision in turn would be expensive; instead perturb and re-solve)."""
    from scipy.optimize import linear_sum_assignment
    nr, nc = cost.shape
    n = nr + nc
    ext = np.full((n, n), limit / 2.0)
    ext[nr:, nc:] = 0
    ext[:nr, :nc] = cost
    r, c = linear_sum_assignment(ext)
    best = ext[r, c].sum()
    # forbid each matched real pair in turn: if the optimum does not get strictly worse there is a tie
    for i in range(nr):
        if x[i] >= 0:
            e2 = ext.copy()
            e2[i, x[i]] = 1e9
            r2, c2 = linear_sum_assignment(e2)
            if e2[r2, c2].sum() <= best + 1e-12:
                return False
    return True
