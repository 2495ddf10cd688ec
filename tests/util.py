"""shared helpers of the test-suite (golden fixture loading, track-stream comparison)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRACKER_CASES = ["sort_default", "bytetrack_default", "bytetrack_default_gaps", "bytetrack_botsort", "sort_strongsort",
                 "bytetrack_crowd", "botsort_gmc", "botsort_crowd"]
DEEPSORT_CASES = ["deepsort_default", "deepsort_crowd", "deepsort_dim512", "deepsort_dim100", "deepsort_identity", "deepsort_identity512"]      # appearance features from synth.make_features at the get_feature seam
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
    """the appearance features a DeepSORT case was recorded with, as a function of the boxes (the get_feature seam)"""
    from yolov7_tracker_amd import synth
    g = np.load(os.path.join(GOLDEN, "tracker_%s.npz" % name))
    dim = tracker_feat_dim(name)
    if "feat_kind" in g.files and str(g["feat_kind"]) == "identity":
        nf, nobj, seq = (int(v) for v in g["scene"])
        dets, fn = synth.make_identity_features(nf, nobj, 1280, seq_idx=seq, dim=dim, miss=float(g["feat_miss"]))
        assert np.array_equal(np.concatenate(dets, 0), g["dets"]), "the regenerated scene is not the recorded one"
        return fn
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


def tie_prone_iou_costs(rng, n_cases=300):
    """IoU cost matrices of INTEGER boxes on a coarse lattice: IoUs are small rationals, so pairs exactly at the limit (1 - IoU == 0.5 / 0.6) and
    equal-cost alternatives are common -- the optimum of linear_assignment is then not unique and the answer is whatever lapjv's algorithm returns"""
    from oracle import cnative
    out = []
    for t in range(n_cases):
        nr, nc = (int(v) for v in rng.integers(1, 14, 2))
        def boxes(n):
            xy = rng.integers(0, 8, (n, 2)) * 10.0
            wh = rng.choice([19.0, 29.0, 39.0], (n, 2))
            return np.concatenate([xy, xy + wh], 1)
        a, b = boxes(nr), boxes(nc)
        out.append((1.0 - cnative.bbox_overlaps(a, b), [0.5, 0.6, 0.9, 0.7][t % 4]))
    return out


def sort_tie_scenes():
    """two random SORT scenes in which a frame's assignment has a tie (found by sweeping the seeds of the random-scene tests): (seed, scene index)"""
    return [(5, 6), (58, 7)]


def random_scene(seed, scene_index, kind="sort"):
    """scene number `scene_index` of the random-scene generator seeded with `seed` (tests/test_hostsim.py)"""
    from yolov7_tracker_amd import synth
    rng = np.random.default_rng(seed)
    for scene in range(scene_index + 1):
        n_obj = int(rng.integers(5, 120))
        n_frames = int(rng.integers(15, 40))
        miss, fp = float(rng.uniform(0.0, 0.3)), float(rng.uniform(0.0, 0.2))
        gap = int(rng.integers(0, 9))
    dets = synth.make_detections(n_frames, n_obj, 640, seq_idx=100 + scene_index, miss=miss, fp=fp)
    if gap > 2:
        dets = [None if (i % gap == gap - 1) else d for i, d in enumerate(dets)]
    if scene_index % 4 == 3:
        dets[n_frames // 2] = np.zeros((0, 6), np.float32)
    return dets


def random_deepsort_scene(seed):
    """-> (dets per frame, feature_fn, dim): 5..90 objects, 15..45 frames, random miss / clutter rates, feature dimension and frame gaps; even seeds
    use box-size features (many near-identical vectors: appearance ties and near-ties), odd seeds identity features with noise"""
    from yolov7_tracker_amd import synth
    rng = np.random.default_rng(seed)
    n_obj, n_frames = int(rng.integers(5, 90)), int(rng.integers(15, 45))
    miss, fp, dim = float(rng.uniform(0.0, 0.35)), float(rng.uniform(0.0, 0.2)), int(rng.choice([32, 100, 128, 512]))
    if seed % 2:
        dets, fn = synth.make_identity_features(n_frames, n_obj, 640, seq_idx=300 + seed, dim=dim, miss=miss, fp=fp, noise=float(rng.uniform(0.05, 0.5)))
    else:
        dets = synth.make_detections(n_frames, n_obj, 640, seq_idx=300 + seed, miss=miss, fp=fp)
        fn = lambda b, _d=dim: synth.make_features(b, dim=_d)
    gap = int(rng.integers(0, 9))
    if gap > 2:
        dets = [None if (i % gap == gap - 1) else d for i, d in enumerate(dets)]
    return dets, fn, dim


def training_checkpoint_state_dict(spec, plan, seed=0, bn_bias_mean=0.0, calib_image=None):
    """a state dict shaped like what the reference's train_aux.py saves for cfg/training/yolov7-w6.yaml (models/yolo.py:111-158): the live layers of
    the plan (seeded, BatchNorm calibrated) + ImplicitA / ImplicitM of the main head (non-trivial values, so that a missed fold shows) + the parameters
    of the aux branch (convs 118-121, m2 / ia-less aux Detect convs) that inference computes and throws away -- the product must ignore them."""
    import torch
    from yolov7_tracker_amd.detector import graph, weights
    nodes, _ = graph.parse(spec)
    sd = weights.calibrate_bn(nodes, weights.random_state_dict(plan.wlayout, seed, bn_bias_mean=bn_bias_mean), seed=seed, image=calib_image)
    det = next(n for n in nodes if n.kind == "detect")
    g = torch.Generator().manual_seed(1000 + seed)
    base = "model.%d" % det.layer
    na_no = det.extra["na"] * det.extra["no"]
    for l, cin in enumerate(det.extra["cin"]):
        sd["%s.ia.%d.implicit" % (base, l)] = torch.randn((1, cin, 1, 1), generator=g) * 0.3          # ImplicitA: x + a   (common.py:433-443)
        sd["%s.im.%d.implicit" % (base, l)] = 1.0 + torch.randn((1, na_no, 1, 1), generator=g) * 0.2   # ImplicitM: x * m   (common.py:446-456)
    # dead parameters of the aux branch (yolo.py:121,141-153): aux convs 118..121 (Conv+BN) and the m2 1x1 convs
    aux_c = {118: (320, 128), 119: (640, 256), 120: (960, 384), 121: (1280, 512)}
    for l, (layer, (c2, c1)) in enumerate(sorted(aux_c.items())):
        sd["model.%d.conv.weight" % layer] = torch.randn((c2, c1, 3, 3), generator=g) * 0.01
        for k, v in (("weight", 1.0), ("bias", 0.0), ("running_mean", 0.0), ("running_var", 1.0)):
            sd["model.%d.bn.%s" % (layer, k)] = torch.full((c2,), v)
        sd["%s.m2.%d.weight" % (base, l)] = torch.randn((na_no, c2, 1, 1), generator=g) * 0.01
        sd["%s.m2.%d.bias" % (base, l)] = torch.zeros(na_no)
    return sd


from oracle.detector_torch import candidates as oracle_candidates, compare_candidate_sets  # noqa: E402,F401  (test-side names)


def lattice_scene(n_frames=8, nx=40, ny=4, seed=0, extra_cols=0):
    """a scene whose association graph is ONE connected component far larger than a wave: nx x ny boxes of 100 x 100 px on a 30 x 60 px lattice (every box overlaps
    its neighbours two to the left / right and one up / down at IoU >= 0.1, i.e. ~11 candidate edges per track -- under the sparse solver's 24 per row), jittered by
    +-3 px (sub-pixel) per frame so that no two costs tie.  extra_cols: that many additional detections per frame half a pitch off the lattice (unmatched high-score boxes: more
    columns than rows).  -> list of (n, 6) float32 [x1, y1, x2, y2, conf, cls]"""
    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.arange(nx) * 30.0, np.arange(ny) * 60.0)
    base = np.stack([gx.ravel(), gy.ravel()], 1) + 20.0
    out = []
    for f in range(n_frames):
        xy = base + rng.uniform(-3, 3, base.shape) + f * 1.5
        if extra_cols and f > 0:
            more = base[rng.choice(len(base), extra_cols, replace=False)] + np.array([15.0, 30.0]) + rng.uniform(-3, 3, (extra_cols, 2)) + f * 1.5
            xy = np.concatenate([xy, more])
        wh = 100.0 + rng.uniform(-2, 2, (len(xy), 2))
        d = np.zeros((len(xy), 6), np.float32)
        d[:, 0:2] = xy; d[:, 2:4] = xy + wh          # (sub-pixel coordinates: integer boxes of one size tie in IoU, and a tie sends the whole problem to the literal solver)
        d[:, 4] = rng.uniform(0.6, 0.95, len(xy)); d[:, 5] = 0
        out.append(d[rng.permutation(len(d))])
    return out
