"""CPU: the oracle restatements (oracle/*.py, oracle/y7t_oracle.c) against the committed golden vectors that
tests/golden/make_golden.py recorded from the REFERENCE's own sources, and against the reference itself when
/root/reference is present."""
import numpy as np
import pytest

from oracle import cnative, tracker_np
from oracle.kalman_np import KalmanNP
from tests import util


@pytest.fixture(scope="module")
def kal():
    return np.load(util.GOLDEN + "/kalman.npz")


@pytest.mark.parametrize("kind", ["default", "botsort", "strongsort"])
def test_kalman_np_matches_reference_golden(kal, kind):
    f = KalmanNP(kind)
    mean, cov, z, conf = (kal[kind + "_" + k] for k in ("mean", "cov", "z", "conf"))
    mp, cp = f.multi_predict(mean, cov)
    np.testing.assert_allclose(mp, kal[kind + "_pred_mean"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cp, kal[kind + "_pred_cov"], rtol=1e-11, atol=1e-11)
    for i in range(len(mean)):
        c = conf[i] if kind == "strongsort" else 0.0
        pm, pc = f.project(mean[i], cov[i], c)
        np.testing.assert_allclose(pm, kal[kind + "_proj_mean"][i], rtol=1e-12)
        np.testing.assert_allclose(pc, kal[kind + "_proj_cov"][i], rtol=1e-12)
        um, uc = f.update(mean[i], cov[i], z[i], c)
        np.testing.assert_allclose(um, kal[kind + "_upd_mean"][i], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(uc, kal[kind + "_upd_cov"][i], rtol=1e-8, atol=1e-9)
        a, b = f.initiate(z[i].astype(np.float32), f32_std=True)
        np.testing.assert_allclose(a, kal[kind + "_init32_mean"][i], rtol=0, atol=0)
        np.testing.assert_allclose(b, kal[kind + "_init32_cov"][i], rtol=1e-7)
        a, b = f.initiate(z[i])
        np.testing.assert_allclose(a, kal[kind + "_init64_mean"][i], rtol=0, atol=0)
        np.testing.assert_allclose(b, kal[kind + "_init64_cov"][i], rtol=1e-14)
        if kind != "botsort":
            np.testing.assert_allclose(f.gating_distance(mean[i], cov[i], z), kal[kind + "_gate4"][i], rtol=1e-9)
            np.testing.assert_allclose(f.gating_distance(mean[i], cov[i], z, True), kal[kind + "_gate2"][i], rtol=1e-9)


@pytest.mark.parametrize("name", util.TRACKER_CASES + util.ORACLE_ONLY_CASES)
def test_tracker_np_matches_reference_golden(name):
    trk, fmt, dets, want = util.load_tracker_case(name)
    from yolov7_tracker_amd import synth
    got = tracker_np.run(trk, dets, kalman_format=fmt, warps=util.load_tracker_warps(name),
                         feature_fn=util.feature_fn_for(name) if trk == "deepsort" else None)
    util.assert_same_tracks(got, want, name)


def test_lap_iou_golden_and_scipy():
    g = np.load(util.GOLDEN + "/lap_iou.npz")
    for k in range(int(g["n_cases"])):
        a, b = g["a%d" % k], g["b%d" % k]
        iou = cnative.bbox_overlaps(a, b)
        np.testing.assert_array_equal(iou, g["iou%d" % k])
        _, x, y = cnative.lapjv(1 - iou, extend_cost=True, cost_limit=float(g["lim%d" % k]))
        np.testing.assert_array_equal(x, g["x%d" % k])
        np.testing.assert_array_equal(y, g["y%d" % k])


def test_bbox_overlaps_bruteforce():
    rng = np.random.default_rng(3)
    a = rng.uniform(0, 100, (17, 4)); a[:, 2:] += a[:, :2]
    b = rng.uniform(0, 100, (9, 4)); b[:, 2:] += b[:, :2]
    got = cnative.bbox_overlaps(a, b)
    for i in range(17):
        for j in range(9):
            iw = min(a[i, 2], b[j, 2]) - max(a[i, 0], b[j, 0]) + 1
            ih = min(a[i, 3], b[j, 3]) - max(a[i, 1], b[j, 1]) + 1
            want = 0.0
            if iw > 0 and ih > 0:
                ua = (a[i, 2] - a[i, 0] + 1) * (a[i, 3] - a[i, 1] + 1) + (b[j, 2] - b[j, 0] + 1) * (b[j, 3] - b[j, 1] + 1) - iw * ih
                want = iw * ih / ua
            assert got[i, j] == want


def test_lapjv_total_cost_is_optimal_vs_scipy():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(5)
    for t in range(60):
        nr, nc = rng.integers(1, 30, 2)
        c = rng.random((nr, nc))
        lim = [0.9, 0.5, 0.7][t % 3]
        _, x, y = cnative.lapjv(c, extend_cost=True, cost_limit=lim)
        n = nr + nc
        ext = np.full((n, n), lim / 2); ext[nr:, nc:] = 0; ext[:nr, :nc] = c
        r, cc = linear_sum_assignment(ext)
        tot = sum(c[i, x[i]] for i in range(nr) if x[i] >= 0) + lim / 2 * ((x < 0).sum() + (y < 0).sum())
        assert abs(tot - ext[r, cc].sum()) < 1e-9
        for i in range(nr):
            if x[i] >= 0:
                assert y[x[i]] == i and c[i, x[i]] < lim


def test_oracle_pinned_against_live_reference(have_reference):
    if not have_reference:
        pytest.skip("/root/reference not present (GPU box); golden vectors cover this")
    from oracle import ref_harness
    from yolov7_tracker_amd import synth
    dets = synth.make_detections(40, 50, seq_idx=9)
    for trk in ("sort", "bytetrack"):
        ref = ref_harness.run_reference_tracker(trk, dets)
        got = tracker_np.run(trk, dets)
        util.assert_same_tracks(got, ref, trk)


@pytest.mark.parametrize("kind,fmt", [("sort", "default"), ("bytetrack", "default"), ("bytetrack", "strongsort"), ("botsort", "botsort"), ("deepsort", "default")])
def test_oracle_equals_live_reference_on_random_scenes(have_reference, kind, fmt):
    """the scenes of the random-scene tests (tests/test_hostsim.py: density, misses, clutter, gaps, empty frames, warps; DeepSORT: the seeds with
    appearance ties and near-ties among them) through the reference's OWN tracker classes and through the oracle: every id and box equal.  This is
    what lets the device tests use the oracle where /root/reference is absent"""
    if not have_reference:
        pytest.skip("/root/reference not present (GPU box)")
    import zlib
    from oracle import ref_harness
    from yolov7_tracker_amd import synth
    if kind == "deepsort":
        for seed in (20, 50, 62, 76, 1, 2, 3):
            dets, fn, dim = util.random_deepsort_scene(seed)
            ref = ref_harness.run_reference_tracker("deepsort", dets, feature_fn=fn)
            util.assert_same_tracks(tracker_np.run("deepsort", dets, feature_fn=fn), ref, "deepsort seed %d (dim %d)" % (seed, dim))
        return
    rng = np.random.default_rng(zlib.crc32(("%s/%s" % (kind, fmt)).encode()))
    for scene in range(8):
        n_obj, n_frames = int(rng.integers(5, 120)), int(rng.integers(15, 40))
        dets = synth.make_detections(n_frames, n_obj, 640, seq_idx=100 + scene, miss=float(rng.uniform(0.0, 0.3)), fp=float(rng.uniform(0.0, 0.2)))
        gap = int(rng.integers(0, 9))
        if gap > 2:
            dets = [None if (i % gap == gap - 1) else d for i, d in enumerate(dets)]
        if scene % 4 == 3:
            dets[n_frames // 2] = np.zeros((0, 6), np.float32)
        warps = synth.make_warps(n_frames, seq_idx=scene) if kind == "botsort" else None
        ref = ref_harness.run_reference_tracker(kind, dets, opts=ref_harness.make_opts(kalman_format=fmt), warps=warps)
        util.assert_same_tracks(tracker_np.run(kind, dets, kalman_format=fmt, warps=warps), ref, "%s/%s scene %d" % (kind, fmt, scene))
