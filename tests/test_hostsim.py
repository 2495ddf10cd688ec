"""CPU: the portable workgroup programs of the device tracker (csrc/y7t_track_core.h, y7t_track_step.h) compiled
for the host with one thread (tests/_hostsim), against the reference's golden vectors and the oracle.  This checks
the control flow / arithmetic of the text that hipcc compiles for gfx950; the `-m gpu` tests check the real thing."""
import ctypes

import numpy as np
import pytest

from oracle import cnative
from tests import _hostsim as hs
from tests import util


@pytest.mark.parametrize("name", util.TRACKER_CASES)
def test_hostsim_tracker_matches_reference_golden(name):
    trk, fmt, dets, want = util.load_tracker_case(name)
    got = hs.run(trk, dets, kalman_format=fmt, warps=util.load_tracker_warps(name))
    util.assert_same_tracks(got, want, name)


@pytest.mark.parametrize("name", util.TRACKER_CASES)
@pytest.mark.parametrize("frames", [1, 7, 32])
def test_hostsim_tracker_with_launch_long_list_arena_matches_reference_golden(name, frames):
    """the frames of one k_tracker_step_frames launch keep the pool's index lists in workgroup memory (y7t_arena_load / y7t_arena_store around them): same
    tracks as the reference, for launches of 1, 7 and 32 frames (the arena is poisoned before every load)"""
    trk, fmt, dets, want = util.load_tracker_case(name)
    got = hs.run(trk, dets, kalman_format=fmt, warps=util.load_tracker_warps(name), arena_frames=frames)
    util.assert_same_tracks(got, want, "%s, arena over %d frames" % (name, frames))


def test_hostsim_list_arena_with_unequal_capacities():
    """the arena's layout follows the pool's capacities (13 track-sized lists, 3 detection-sized ones, one of the larger size): a pool with fewer track slots than
    detection slots and one with more, against the same golden sequence"""
    name = util.TRACKER_CASES[0]
    trk, fmt, dets, want = util.load_tracker_case(name)
    for cap_t, cap_d in ((256, 640), (768, 192)):
        got = hs.run(trk, dets, kalman_format=fmt, warps=util.load_tracker_warps(name), arena_frames=4, cap_t=cap_t, cap_d=cap_d)
        util.assert_same_tracks(got, want, "%s, arena, capacities %d x %d" % (name, cap_t, cap_d))


@pytest.mark.parametrize("name", util.DEEPSORT_CASES)
def test_hostsim_deepsort_matches_reference_golden(name):
    """DeepSORT's workgroup program (csrc/y7t_track_deepsort.h): matching cascade over the gated appearance cost, the IoU fallbacks, the
    reference's two index quirks -- ids AND boxes equal to the sequences recorded from the reference's deepsort.py"""
    from yolov7_tracker_amd import synth
    trk, fmt, dets, want = util.load_tracker_case(name)
    got = hs.run(trk, dets, kalman_format=fmt, feature_fn=util.feature_fn_for(name), feat_dim=util.tracker_feat_dim(name))
    util.assert_same_tracks(got, want, name)


def test_hostsim_deepsort_joint_cascade_is_exercised():
    """with features that identify the object no detection is wanted by tracks of two ages, and the step solves all cascade levels in one
    assignment -- the identity goldens must actually take that path (and the size-feature goldens the level-by-level one)"""
    L = hs.lib()
    for name, joint in (("deepsort_identity", True), ("deepsort_default", False)):
        trk, fmt, dets, want = util.load_tracker_case(name)
        c0 = L.hs_tie_reason(7)
        got = hs.run(trk, dets, kalman_format=fmt, feature_fn=util.feature_fn_for(name), feat_dim=util.tracker_feat_dim(name))
        util.assert_same_tracks(got, want, name)
        n = L.hs_tie_reason(7) - c0
        assert (n > len(dets) // 2) if joint else (n == 0), (name, n)


def test_hostsim_cpython_set_order_emulation():
    """matching_cascade returns list(set(range(n)) - set(matched)) (matching.py:275); its ORDER is CPython's set-table order and decides
    which tracks deepsort.py:171-173 marks lost -- the emulation must agree with the real thing"""
    import random
    rnd = random.Random(0)
    for trial in range(1500):
        n = rnd.choice([1, 2, 5, 8, 19, 20, 33, 77, 78, 100, 150, 307, 308, 400, 700, 1000, 1200]) if trial % 3 else rnd.randint(0, 1228)
        m = rnd.sample(range(n), rnd.randint(0, n)) if n else []
        assert hs.pyset_difference(n, m) == list(set(range(n)) - set(k for k in m)), (n, len(m))


def test_hostsim_lapjv_equals_oracle():
    rng = np.random.default_rng(0)
    for t in range(200):
        nr, nc = rng.integers(1, 45, 2)
        c = rng.random((nr, nc))
        if t % 3 == 0:
            c = np.where(rng.random((nr, nc)) < 0.7, 1.0, c)  # mostly "no overlap", like real IoU costs
        lim = [0.9, 0.5, 0.7][t % 3]
        _, x0, y0 = cnative.lapjv(c, extend_cost=True, cost_limit=lim)
        x1, y1 = hs.lapjv(c, lim)
        np.testing.assert_array_equal(x0, x1)
        np.testing.assert_array_equal(y0, y1)


def test_hostsim_reduced_sap_solver_equals_oracle():
    """the reduced shortest-augmenting-path solver (the one the device uses) returns lap's assignment on problems with a
    unique optimum -- random costs, IoU-like sparse costs, degenerate shapes"""
    rng = np.random.default_rng(1)
    for t in range(400):
        nr, nc = rng.integers(1, 60, 2)
        c = rng.random((nr, nc))
        if t % 3 == 0:
            c = np.where(rng.random((nr, nc)) < 0.8, 1.0, c)
        if t % 7 == 0:
            c = c * 0.3          # everything below the limit: dense competition
        lim = [0.9, 0.5, 0.7][t % 3]
        _, x0, y0 = cnative.lapjv(c, extend_cost=True, cost_limit=lim)
        x1, y1 = hs.lapjv(c, lim, sap=True)
        np.testing.assert_array_equal(x0, x1)
        np.testing.assert_array_equal(y0, y1)
    g = np.load(util.GOLDEN + "/lap_iou.npz")
    for k in range(int(g["n_cases"])):
        x1, y1 = hs.lapjv(1 - g["iou%d" % k], float(g["lim%d" % k]), sap=True)
        np.testing.assert_array_equal(x1, g["x%d" % k])
        np.testing.assert_array_equal(y1, g["y%d" % k])


def test_hostsim_assignment_with_ties_follows_lapjv():
    """non-unique optima (a pair exactly at the limit, equal-cost alternatives): the fast solver notices and re-solves with the literal JV, so the
    assignment is lapjv's in those cases too"""
    n_tied = 0
    for c, lim in util.tie_prone_iou_costs(np.random.default_rng(3)):
        _, x0, y0 = cnative.lapjv(c, extend_cost=True, cost_limit=lim)
        x1, y1 = hs.lapjv(c, lim, sap=True)
        np.testing.assert_array_equal(x0, x1)
        np.testing.assert_array_equal(y0, y1)
        n_tied += int((c == lim).any())
    assert n_tied > 20


@pytest.mark.parametrize("seed,scene", util.sort_tie_scenes())
def test_hostsim_sort_scenes_with_assignment_ties(seed, scene):
    from oracle import tracker_np
    dets = util.random_scene(seed, scene)
    util.assert_same_tracks(hs.run("sort", dets, kalman_format="default"), tracker_np.run("sort", dets, kalman_format="default"), "seed %d scene %d" % (seed, scene))


@pytest.mark.parametrize("seeds", [(20, 50, 62, 76), tuple(range(0, 6)), tuple(range(6, 12))])
def test_hostsim_deepsort_equals_oracle_on_random_scenes(seeds):
    """random scenes through DeepSORT (util.random_deepsort_scene): device program == oracle on every id and box.  Box-size features make many
    appearance vectors equal or nearly so, and then the cascade is decided by (a) which of several optimal assignments lapjv returns -- the step
    notices duplicate candidate costs and runs lapjv.cpp literally (seeds 20, 62) -- and (b) the last bit of a cosine distance, which in numpy
    depends on the summation order the BLAS picks for the operands' shape; the oracle is run with the order pinned (tracker_np.dot_sequential, the
    order the kernels use; seeds 20, 50, 62, 76 differ from np.dot's own answer on this machine by exactly such one-ulp near-ties).  The goldens
    above, recorded from the live reference, hold with np.dot as it is."""
    from oracle import tracker_np
    for seed in seeds:
        dets, fn, dim = util.random_deepsort_scene(seed)
        want = tracker_np.run("deepsort", dets, feature_fn=fn, dot=tracker_np.dot_sequential)
        util.assert_same_tracks(hs.run("deepsort", dets, feature_fn=fn, feat_dim=dim), want, "seed %d (dim %d)" % (seed, dim))


def test_oracle_dot_sequential_is_the_blocked_sgemm_order():
    """for the shapes where OpenBLAS runs its blocked kernel on one K panel (tens of stored rows x tens of detections, K <= 128) np.dot IS the
    sequential chain -- which is why the reference-recorded goldens agree bit for bit; elsewhere the two differ by an ulp or two, never more"""
    from oracle import tracker_np
    rng = np.random.default_rng(0)
    for k, n, dim, same in ((100, 60, 128, True), (64, 40, 32, True), (1, 40, 128, False), (100, 60, 512, False)):
        a, b = rng.standard_normal((k, dim)).astype(np.float32), rng.standard_normal((n, dim)).astype(np.float32)
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        x, y = np.dot(a, b.T), tracker_np.dot_sequential(a, b.T)
        assert np.abs(x - y).max() <= 4 * np.finfo(np.float32).eps, (k, n, dim)
        if same and not np.array_equal(x, y):
            pytest.skip("this machine's BLAS uses another order for (%d, %d, %d)" % (k, n, dim))


def test_hostsim_kalman_matches_reference_golden():
    kal = np.load(util.GOLDEN + "/kalman.npz")
    L = hs.lib()
    for kind, kid in (("default", 0), ("botsort", 2), ("strongsort", 3)):
        mean, cov, z, conf = (kal[kind + "_" + k] for k in ("mean", "cov", "z", "conf"))
        for i in range(len(mean)):
            m, P = mean[i].copy(), cov[i].copy()
            L.hs_kf_predict(kid, m.ctypes.data, P.ctypes.data)
            np.testing.assert_allclose(m, kal[kind + "_pred_mean"][i], rtol=1e-13)
            np.testing.assert_allclose(P, kal[kind + "_pred_cov"][i], rtol=1e-12, atol=1e-12)
            m, P = mean[i].copy(), cov[i].copy()
            L.hs_kf_update(kid, m.ctypes.data, P.ctypes.data, z[i].ctypes.data, ctypes.c_double(conf[i] if kind == "strongsort" else 0.0))
            np.testing.assert_allclose(m, kal[kind + "_upd_mean"][i], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(P, kal[kind + "_upd_cov"][i], rtol=1e-8, atol=1e-9)
            z32 = z[i].astype(np.float32).astype(np.float64)
            m, P = np.zeros(8), np.zeros((8, 8))
            L.hs_kf_initiate(kid, z32.ctypes.data, 1, m.ctypes.data, P.ctypes.data)
            np.testing.assert_array_equal(m, kal[kind + "_init32_mean"][i])
            np.testing.assert_allclose(P, kal[kind + "_init32_cov"][i], rtol=1e-7)
            if kind != "botsort":
                for j in range(len(z)):
                    g = L.hs_kf_gating(kid, mean[i].ctypes.data, cov[i].ctypes.data, z[j].ctypes.data, 0)
                    np.testing.assert_allclose(g, kal[kind + "_gate4"][i][j], rtol=1e-9)


def test_hostsim_capacity_error_is_loud():
    from yolov7_tracker_amd import synth
    dets = synth.make_detections(3, 80, seq_idx=0)
    trk = hs.HostSimTracker("bytetrack", cap_t=16, cap_d=128)
    with pytest.raises(RuntimeError):
        for d in dets:
            trk.update(d)


def test_hostsim_empty_and_ragged_frames():
    trk = hs.HostSimTracker("bytetrack")
    assert trk.update(np.zeros((0, 6), np.float32)) == []
    assert trk.update(None) == []
    one = np.array([[10, 10, 50, 90, 0.9, 3]], np.float32)
    out = trk.update(one)           # frame 3: new track is born unconfirmed (only frame-1 births are active)
    assert out == []
    out = trk.update(one)
    assert [r[0] for r in out] == [1]
    out = trk.update(np.zeros((0, 6), np.float32))
    assert out == []


@pytest.mark.parametrize("kind,fmt", [("sort", "default"), ("bytetrack", "default"), ("bytetrack", "strongsort"), ("botsort", "botsort")])
def test_hostsim_tracker_equals_oracle_on_random_scenes(kind, fmt):
    """many short seeded scenes of random density (5..120 objects), miss / clutter rates, frame gaps and camera warps: the device
    program (host build) and the numpy oracle -- itself pinned to the reference -- must agree on every id, in every frame"""
    from oracle import tracker_np
    from yolov7_tracker_amd import synth
    import zlib
    rng = np.random.default_rng(zlib.crc32(("%s/%s" % (kind, fmt)).encode()))      # (hash() of a str changes from process to process)
    for scene in range(12):
        n_obj = int(rng.integers(5, 120))
        n_frames = int(rng.integers(15, 40))
        dets = synth.make_detections(n_frames, n_obj, 640, seq_idx=100 + scene, miss=float(rng.uniform(0.0, 0.3)), fp=float(rng.uniform(0.0, 0.2)))
        gap = int(rng.integers(0, 9))
        if gap > 2:
            dets = [None if (i % gap == gap - 1) else d for i, d in enumerate(dets)]
        if scene % 4 == 3:
            dets[n_frames // 2] = np.zeros((0, 6), np.float32)          # an empty frame in the middle
        warps = synth.make_warps(n_frames, seq_idx=scene) if kind == "botsort" else None
        want = tracker_np.run(kind, dets, kalman_format=fmt, warps=warps)
        got = hs.run(kind, dets, kalman_format=fmt, warps=warps)
        util.assert_same_tracks(got, want, "%s/%s scene %d (%d objects, %d frames, gap %d)" % (kind, fmt, scene, n_obj, n_frames, gap))
        got = hs.run(kind, dets, kalman_format=fmt, warps=warps, arena_frames=1 + scene % 5)      # the same with the index lists in the launch-long arena
        util.assert_same_tracks(got, want, "%s/%s scene %d, list arena" % (kind, fmt, scene))


def test_hostsim_cfg3_full_size_botsort_equals_oracle():
    """BASELINE configs[2] at full size (300 frames x 500 objects, a camera-motion warp per frame; the scene of `bench.py --workload cfg3`): the host
    build of the device BoT-SORT program against the oracle -- every id, every box.  The device run of the same comparison: tests/test_fullsize_gpu.py"""
    from yolov7_tracker_amd import synth
    dets = synth.make_detections(300, 500, 1280, seq_idx=0, bounce=True)
    warps = synth.make_warps(300, seq_idx=0)
    from oracle import tracker_np
    want = tracker_np.run("botsort", dets, kalman_format="botsort", warps=warps)
    before = [hs.lib().hs_next_stat(k) for k in range(4)]
    got = hs.run("botsort", dets, warps=warps, kalman_format="botsort", cap_t=2048, cap_d=1024)
    util.assert_same_tracks(got, want, "cfg3 full size")
    # the large connected components of the crowded frames went through the register-resident wave solve (y7t_assoc_sparse_try step 4a, the host build runs its
    # text with 64-element arrays), none had more than 64 rows or columns
    on_wave, declined = hs.lib().hs_next_stat(2) - before[2], hs.lib().hs_next_stat(3) - before[3]
    assert on_wave > 300 and declined == 0, (on_wave, declined)


def test_hostsim_iou_pretest_never_rejects_an_overlapping_pair():
    """y7t_box_apart (float32, column box grown by 2 px) may only say "apart" where matching.iou_distance's formula gives exactly 1 (iw <= 0 or ih <= 0 with its
    +1 pixel convention); and the distance the cost / duplicate passes use (pre-test, then the exact formula) equals the plain formula bit for bit -- random
    boxes, boxes that touch at 0 / 1 / 2 px (+- an ulp), coordinates up to 4e6 (beyond 2^21 the pre-test must stand aside), infinities and NaN"""
    import ctypes
    L = hs.lib()
    L.hs_box_apart.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.hs_box_iou_dist.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.hs_box_iou_dist.restype = ctypes.c_double
    rng = np.random.default_rng(5)
    cases = []
    for scale in (100.0, 1280.0, 5e4, 2e6, 4e6):
        a = rng.uniform(-scale, scale, (400, 4)); a[:, 2:] = a[:, :2] + rng.uniform(0, scale / 8, (400, 2))
        b = a + rng.normal(0, scale / 50, (400, 4))
        cases += list(zip(a, b))
        for gap in (0.0, 1.0, 2.0, 3.0):      # b to the right of / below a at exactly `gap`, one ulp less, one ulp more
            for eps in (0.0, -1.0, 1.0):
                for axis in (0, 1):
                    q = a[:60].copy()
                    shift = a[:60, 2 + axis] - a[:60, axis] + gap
                    q[:, axis] += shift; q[:, 2 + axis] += shift
                    q[:, axis] = np.nextafter(q[:, axis], q[:, axis] + eps) if eps else q[:, axis]
                    cases += list(zip(a[:60], q)) + list(zip(q, a[:60]))
    odd = np.array([[0, 0, 10, 10], [np.inf, 0, np.inf, 5], [np.nan, 0, 4, 4], [-np.inf, -np.inf, np.inf, np.inf], [5, 5, 1, 1], [3e6, 3e6, 3e6 + 4, 3e6 + 4]], float)
    cases += [(x, y) for x in odd for y in odd]
    n_apart = 0
    for b, q in cases:
        b, q = np.ascontiguousarray(b, np.float64), np.ascontiguousarray(q, np.float64)
        want = np.zeros(1)
        L.hs_iou_cost(b.ctypes.data, 1, q.ctypes.data, 1, want.ctypes.data)
        got = L.hs_box_iou_dist(b.ctypes.data, q.ctypes.data)
        assert got == want[0] or (np.isnan(got) and np.isnan(want[0])), (b, q, got, want[0])
        if L.hs_box_apart(b.ctypes.data, q.ctypes.data):
            n_apart += 1
            assert want[0] == 1.0, (b, q, want[0])
    assert n_apart > len(cases) // 10


@pytest.mark.parametrize("kind,extra", [("bytetrack", 0), ("bytetrack", 60), ("botsort", 0)])
def test_hostsim_component_larger_than_a_wave(kind, extra):
    """a 160-track lattice (tests/util.lattice_scene): the association's candidate graph is one connected component of 160 rows (and, with `extra`, of more than 64
    columns beside few enough rows in the later associations) -- more than the 64 slots of the register-resident wave solve: the wave then solves it with its state in
    the work arrays (y7t_assoc_sparse_try step 4a, the `ncl > 64 || nrw > 64` branch, counted by statistic 3) and still returns the oracle's assignment"""
    from oracle import tracker_np
    dets = util.lattice_scene(extra_cols=extra)
    fmt = "botsort" if kind == "botsort" else "default"
    before, lit = hs.lib().hs_next_stat(3), hs.lib().hs_literal_calls()
    want = tracker_np.run(kind, dets, kalman_format=fmt)
    got = hs.run(kind, dets, kalman_format=fmt)
    util.assert_same_tracks(got, want, "lattice %s +%d" % (kind, extra))
    assert len(want[-1]) >= 150
    assert hs.lib().hs_next_stat(3) > before, "the scene was meant to produce a component larger than a wave"
    assert hs.lib().hs_literal_calls() == lit, "the scene was meant to be free of ties (a tie sends the whole problem to the literal solver, whose order-dependent parts run on one thread)"


@pytest.mark.parametrize("n_obj,size", [(150, 480), (250, 640), (400, 640), (400, 480)])
@pytest.mark.parametrize("kind", ["bytetrack", "botsort"])
def test_hostsim_crowded_scenes_through_every_solver_path(kind, n_obj, size):
    """crowds far denser than BASELINE's configs (150 .. 400 objects on a 480 / 640 px frame, 10 % misses and clutter, camera warps for BoT-SORT): the candidate graph of
    the association then has components of every size -- hundreds go through the register-resident wave solve, and from ~250 objects on some have more than 64 rows or
    columns and are solved by a wave with its state in the work arrays (statistic 3 of y7t_assoc_sparse_try).  Every id and box of every frame equal to the oracle's."""
    from oracle import tracker_np
    from yolov7_tracker_amd import synth
    fmt = "botsort" if kind == "botsort" else "default"
    dets = synth.make_detections(14, n_obj, size, seq_idx=300 + n_obj, miss=0.1, fp=0.1)
    warps = synth.make_warps(14, seq_idx=3) if kind == "botsort" else None
    before = [hs.lib().hs_next_stat(k) for k in range(4)]
    want = tracker_np.run(kind, dets, kalman_format=fmt, warps=warps)
    got = hs.run(kind, dets, kalman_format=fmt, warps=warps, cap_t=2048, cap_d=1024)
    util.assert_same_tracks(got, want, "%s, %d objects on %d px" % (kind, n_obj, size))
    on_wave, declined = hs.lib().hs_next_stat(2) - before[2], hs.lib().hs_next_stat(3) - before[3]
    assert on_wave > 0
    if n_obj >= 400:
        assert declined > 0, "the densest scenes were meant to produce components larger than a wave"
