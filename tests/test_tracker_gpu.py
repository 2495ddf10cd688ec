"""GPU parity tests of the tracker half of the hot path, through the C ABI of liby7t.so.
Oracle: the golden vectors recorded from the reference's own sources (tests/golden) and the CPU restatements in
oracle/.  Bars: integer outputs (assignments, track ids) bit-exact; IoU cost bit-exact (same IEEE operations);
Kalman within the stated float64 tolerances."""
import os
import types

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from yolov7_tracker_amd import _lib
    _lib.require_gpu()
    return _lib.load()


def make_opts(**kw):
    o = types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280, iou_thresh=0.5,
                              reid_model_path="", gamma=0.1, min_area=150)
    o.__dict__.update(kw)
    return o


def run_device_tracker(trk, fmt, dets, warps=None, **kw):
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack, BaseTracker
    from yolov7_tracker_amd.tracker.botsort import BoTSORT
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    BaseTrack._count = 0
    cls = {"sort": BaseTracker, "bytetrack": ByteTrack, "botsort": BoTSORT}[trk]
    t = cls(make_opts(kalman_format=fmt, **kw), frame_rate=30)
    out = []
    for fi, d in enumerate(dets):
        if d is None:
            cur = t.update_without_detection(None, None)
        elif trk == "botsort":
            cur = t.update(d, None, warp=None if warps is None else warps[fi])
        else:
            cur = t.update(d, None)
        out.append([(tr.track_id, tr.tlwh, float(tr.cls), float(tr.score)) for tr in cur])
    return out, t


@pytest.mark.parametrize("name", util.TRACKER_CASES)
def test_fused_tracker_matches_reference_golden(name):
    trk, fmt, dets, want = util.load_tracker_case(name)
    got, _ = run_device_tracker(trk, fmt, dets, warps=util.load_tracker_warps(name))
    util.assert_same_tracks(got, want, name)


@pytest.mark.parametrize("threads", [64, 256, 1024])
def test_fused_tracker_thread_counts_agree(threads):
    trk, fmt, dets, want = util.load_tracker_case("bytetrack_default")
    got, _ = run_device_tracker(trk, fmt, dets[:40], tracker_threads=threads)
    util.assert_same_tracks(got, want[:40], "threads=%d" % threads)


def test_fused_tracker_accepts_device_tensor_and_views():
    trk, fmt, dets, want = util.load_tracker_case("bytetrack_default")
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack, TrackState
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    BaseTrack._count = 0
    t = ByteTrack(make_opts(), frame_rate=30)
    for f in range(12):
        cur = t.update(torch.from_numpy(dets[f]).cuda(), None)
        assert [c.track_id for c in cur] == [r[0] for r in want[f]]
    tracked, lost = t.tracked_stracks, t.lost_stracks
    assert all(s.state == TrackState.Tracked for s in tracked)
    assert all(s.state in (TrackState.Lost, TrackState.Removed) for s in lost)
    s = tracked[0]
    assert s.mean.shape == (8,) and s.cov.shape == (8, 8) and s.frame_id == 12
    np.testing.assert_allclose(s.tlbr[2:] - s.tlbr[:2], s.tlwh[2:])
    assert BaseTrack._count == max(r[0] for fr in want[:12] for r in fr) or BaseTrack._count >= len(tracked)
    assert t.frame_id == 12


def test_batched_step_equals_single_trackers(L):
    """y7t_tracker_step_batch: 4 independent sequences step together in ONE launch (one workgroup each, detection counts read
    on the device) and produce exactly what 4 separately stepped trackers produce (BASELINE config 5 inside one GPU)."""
    import ctypes
    from yolov7_tracker_amd import _lib, synth
    nseq, nfr, cap = 4, 25, 256
    seqs = [synth.make_detections(nfr, 40 + 10 * s, seq_idx=20 + s) for s in range(nseq)]
    nbytes = int(L.y7t_tracker_state_bytes(cap, cap))

    def mk():
        st = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(nseq)]
        ids = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(nseq)]      # one id counter per sequence
        for s in range(nseq):
            _lib.check(L.y7t_tracker_init(_lib.ptr(st[s]), nbytes, 1, 0, cap, cap, 0.2, 0.5, 30, 1, _lib.ptr(ids[s]), _lib.stream_ptr()))
        return st, ids
    outs = torch.zeros((nseq, cap + 1, 8), dtype=torch.float64, device="cuda")
    # (a) one tracker at a time
    st, _ = mk()
    single = [[] for _ in range(nseq)]
    for f in range(nfr):
        for s in range(nseq):
            d = torch.from_numpy(seqs[s][f]).cuda()
            _lib.check(L.y7t_tracker_step(_lib.ptr(st[s]), _lib.ptr(d), d.shape[0], _lib.ptr(outs[s]), cap,
                                          ctypes.c_void_p(outs[s].data_ptr() + cap * 64), 0, None, _lib.stream_ptr()))
            torch.cuda.synchronize()
            h = outs[s].cpu().numpy()
            single[s].append(h[:int(h[cap].view(np.int32)[0])].copy())
    # (b) all four in one launch per frame
    st, _ = mk()
    state_ptrs = torch.tensor([t.data_ptr() for t in st], dtype=torch.int64, device="cuda")
    out_ptrs = torch.tensor([outs[s].data_ptr() for s in range(nseq)], dtype=torch.int64, device="cuda")
    counts = torch.zeros(nseq, dtype=torch.int32, device="cuda")
    for f in range(nfr):
        dd = [torch.from_numpy(seqs[s][f]).cuda() for s in range(nseq)]
        det_ptrs = torch.tensor([d.data_ptr() for d in dd], dtype=torch.int64, device="cuda")
        n_dev = torch.tensor([d.shape[0] for d in dd], dtype=torch.int32, device="cuda")
        _lib.check(L.y7t_tracker_step_batch(_lib.ptr(state_ptrs), _lib.ptr(det_ptrs), _lib.ptr(n_dev), _lib.ptr(out_ptrs), _lib.ptr(counts), cap,
                                            nseq, 0, None, _lib.stream_ptr()))
        torch.cuda.synchronize()
        c = counts.cpu().numpy()
        h = outs.cpu().numpy()
        for s in range(nseq):
            np.testing.assert_array_equal(h[s, :c[s]], single[s][f])
            assert c[s] == len(single[s][f])


def test_batched_step_with_one_shared_id_counter(L):
    """BaseTrack._count is ONE counter for every tracker of the process (basetrack.py:22,43-46).  8 trackers stepped concurrently by
    y7t_tracker_step_batch on the same counter: no id is handed out twice or skipped, every tracker's ids of one frame are a
    consecutive block (the frame's activate() calls in order), and the counter ends at the number of tracks ever created."""
    from yolov7_tracker_amd import _lib, synth
    nseq, nfr, cap = 8, 20, 256
    seqs = [synth.make_detections(nfr, 60, seq_idx=40 + s) for s in range(nseq)]
    nbytes = int(L.y7t_tracker_state_bytes(cap, cap))
    idc = torch.zeros(1, dtype=torch.int32, device="cuda")
    st = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(nseq)]
    for s in range(nseq):
        _lib.check(L.y7t_tracker_init(_lib.ptr(st[s]), nbytes, 1, 0, cap, cap, 0.2, 0.5, 30, 1, _lib.ptr(idc), _lib.stream_ptr()))
    outs = torch.zeros((nseq, cap + 1, 8), dtype=torch.float64, device="cuda")
    state_ptrs = torch.tensor([t.data_ptr() for t in st], dtype=torch.int64, device="cuda")
    out_ptrs = torch.tensor([outs[s].data_ptr() for s in range(nseq)], dtype=torch.int64, device="cuda")
    counts = torch.zeros(nseq, dtype=torch.int32, device="cuda")
    lay_n = L.y7t_tracker_layout(cap, cap, None, 0)
    import ctypes
    offs = (ctypes.c_int64 * lay_n)()
    L.y7t_tracker_layout(cap, cap, offs, lay_n)
    lay = {L.y7t_tracker_field_name(i).decode(): int(offs[i]) for i in range(lay_n)}
    seen, n_new = {}, 0
    for f in range(nfr):
        before = int(idc.item())
        dd = [torch.from_numpy(seqs[s][f]).cuda() for s in range(nseq)]
        det_ptrs = torch.tensor([d.data_ptr() for d in dd], dtype=torch.int64, device="cuda")
        n_dev = torch.tensor([d.shape[0] for d in dd], dtype=torch.int32, device="cuda")
        _lib.check(L.y7t_tracker_step_batch(_lib.ptr(state_ptrs), _lib.ptr(det_ptrs), _lib.ptr(n_dev), _lib.ptr(out_ptrs), _lib.ptr(counts), cap,
                                            nseq, 0, None, _lib.stream_ptr()))
        torch.cuda.synchronize()
        after = int(idc.item())
        spans = []
        for s in range(nseq):
            raw = st[s].cpu().numpy()
            tid = raw[lay["tid"]:lay["tid"] + 4 * cap].view(np.int32)
            start = raw[lay["start"]:lay["start"] + 4 * cap].view(np.int32)
            mine = sorted(int(t) for t, sf in zip(tid, start) if t > before and sf == f + 1)   # born this frame, still in the pool
            for t in mine:
                assert before < t <= after and t not in seen, "id %d handed out twice / out of range" % t
                seen[t] = s
            if mine:
                spans.append((mine[0], mine[-1], s))
        spans.sort()
        for a, b in zip(spans, spans[1:]):
            assert a[1] < b[0], "id blocks of trackers %d and %d interleave: %r %r" % (a[2], b[2], a, b)   # one block per tracker and frame
        n_new += after - before
    assert int(idc.item()) == n_new >= len(seen) > nseq * 40


@pytest.mark.parametrize("name", util.DEEPSORT_CASES)
def test_device_deepsort_matches_reference_golden(name):
    """DeepSORT on the device (tracker kind 3; features injected at the reference's get_feature seam): ids and boxes of every frame equal
    to the sequences recorded from the reference's deepsort.py (matching cascade, gate_cost_matrix, nearest_embedding_distance)"""
    import types
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.deepsort import DeepSORT
    trk, fmt, dets, want = util.load_tracker_case(name)
    assert trk == "deepsort"
    BaseTrack._count = 0
    t = DeepSORT(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format=fmt, img_size=1280, iou_thresh=0.5))
    feat = util.feature_fn_for(name)          # 128 wide, 512 (OSNet's width: wave-parallel norms, tiled distance kernel), 100 (the plain forms)
    t.get_feature = lambda tlbrs, img: feat(tlbrs)
    got = []
    for d in dets:
        cur = t.update_without_detection() if d is None else t.update(d, None)
        got.append([(c.track_id, c.tlwh, float(c.cls), float(c.score)) for c in cur])
    util.assert_same_tracks(got, want, name)
    assert len(t.tracked_stracks) > 0 and t.frame_id == len(dets)


def test_device_deepsort_chain_without_host_round_trips():
    """DeepSORT._launch (what bench.py --workload cfg4 enqueues: detections and features already on the device, results left on the device, no
    host sync between frames) reproduces the reference-recorded sequence row for row"""
    import types
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.deepsort import DeepSORT
    name = "deepsort_dim512"
    trk, fmt, dets, want = util.load_tracker_case(name)
    assert all(d is not None for d in dets)
    feat = util.feature_fn_for(name)
    BaseTrack._count = 0
    t = DeepSORT(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format=fmt, img_size=1280, iou_thresh=0.5))
    ddev = [torch.from_numpy(d).cuda() for d in dets]
    fdev = [torch.from_numpy(feat(d[:, :4])).cuda() for d in dets]
    res = torch.zeros((len(dets), t.cap_t + 1, 8), dtype=torch.float64, device="cuda")
    for i in range(len(dets)):
        t._launch(ddev[i], fdev[i], out=res[i])
    torch.cuda.synchronize()
    res = res.cpu().numpy()
    got = []
    for i in range(len(dets)):
        cnt = int(res[i, t.cap_t].view(np.int32)[0])
        got.append([(int(r[0]), r[1:5], float(np.float32(r[5])), float(np.float32(r[6]))) for r in res[i, :cnt]])
    util.assert_same_tracks(got, want, name)


def test_deepsort_crops_reach_the_reid_callable():
    """the default get_feature crops ori_img like deepsort.py:28-34 and hands the crops to reid_model"""
    import types
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.deepsort import DeepSORT
    seen = []

    def reid(crops):
        seen.append([c.shape for c in crops])
        return np.stack([np.full(16, float(c.shape[0] * 100 + c.shape[1]), np.float32) + np.arange(16, dtype=np.float32) for c in crops])
    BaseTrack._count = 0
    t = DeepSORT(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=640, iou_thresh=0.5), reid_model=reid)
    img = np.zeros((480, 640, 3), np.uint8)
    det = np.array([[10, 20, 50, 100, 0.9, 0], [200, 100, 260, 220, 0.8, 1], [300, 300, 320, 340, 0.1, 0]], np.float32)
    cur = t.update(det, img)
    assert seen == [[(80, 40, 3), (120, 60, 3)]] and [c.track_id for c in cur] == [1, 2]
    cur = t.update(det + np.float32([2, 1, 2, 1, 0, 0]), img)
    assert [c.track_id for c in cur] == [1, 2]


def test_plugin_tracker_written_against_the_strack_contract():
    """a tracker plugin written like the reference's own trackers -- per-object STrack.activate / update / re_activate /
    multi_predict, matching.iou_distance / linear_assignment, joint / sub / remove_duplicate_stracks -- runs on this package's classes (the
    arithmetic in liby7t.so's per-op kernels) and reproduces the reference's SORT golden sequence id for id."""
    from yolov7_tracker_amd.tracker import matching
    from yolov7_tracker_amd.tracker.basetrack import (BaseTrack, STrack, TrackState, KALMAN_DICT, joint_stracks, sub_stracks,
                                                      remove_duplicate_stracks)
    trk, fmt, dets, want = util.load_tracker_case("sort_default")
    n_frames = 25

    class PluginSORT:                      # the control flow of basetrack.py:368-487, restated against the public surface only
        def __init__(self):
            self.tracked, self.lost, self.removed, self.frame_id = [], [], [], 0
            self.kalman = KALMAN_DICT[fmt]()

        def update(self, det):
            self.frame_id += 1
            act, refind, lost, removed = [], [], [], []
            det = det[det[:, 4] > 0.2]
            D = [STrack(c, STrack.tlbr2tlwh(b), s_, kalman_format=fmt) for c, b, s_ in zip(det[:, 5], det[:, :4], det[:, 4])]
            unconf = [t for t in self.tracked if not t.is_activated]
            pool = joint_stracks([t for t in self.tracked if t.is_activated], self.lost)
            STrack.multi_predict(pool, self.kalman)
            m, ut, ud = matching.linear_assignment(matching.iou_distance(pool, D), thresh=0.5)
            for it, idt in m:
                t = pool[it]
                if t.state == TrackState.Tracked:
                    t.update(D[idt], self.frame_id); act.append(t)
                else:
                    t.re_activate(D[idt], self.frame_id, new_id=False); refind.append(t)
            for it in ut:
                if pool[it].state == TrackState.Tracked:
                    pool[it].mark_lost(); lost.append(pool[it])
            left = [D[i] for i in ud]
            m, ut, ud = matching.linear_assignment(matching.iou_distance(unconf, left), thresh=0.6)
            for it, idt in m:
                t = unconf[it]
                if t.state == TrackState.Tracked:
                    t.update(left[idt], self.frame_id); act.append(t)
                else:
                    t.re_activate(left[idt], self.frame_id, new_id=False); refind.append(t)
            for it in ut:
                unconf[it].mark_removed(); removed.append(unconf[it])
            for i in ud:
                if left[i].score > 0.2 + 0.1:
                    left[i].activate(self.frame_id); act.append(left[i])
            for t in self.lost:
                if self.frame_id - t.end_frame > 30:
                    t.mark_removed(); removed.append(t)
            self.tracked = [t for t in self.tracked if t.state == TrackState.Tracked]
            self.tracked = joint_stracks(joint_stracks(self.tracked, act), refind)
            self.lost = sub_stracks(self.lost, self.tracked)
            self.lost.extend(lost)
            self.lost = sub_stracks(self.lost, self.removed)
            self.removed.extend(removed)
            self.tracked, self.lost = remove_duplicate_stracks(self.tracked, self.lost)
            return [t for t in self.tracked if t.is_activated]

    BaseTrack._count = 0
    p = PluginSORT()
    for f in range(n_frames):
        cur = p.update(np.asarray(dets[f], dtype=np.float32))
        assert [t.track_id for t in cur] == [r[0] for r in want[f]], f
        for t, r in zip(cur, want[f]):
            np.testing.assert_allclose(np.asarray(t.tlwh, dtype=np.float64), r[1], rtol=util.TLWH_RTOL, atol=util.TLWH_ATOL)
    assert BaseTrack._count >= max(r[0] for fr in want[:n_frames] for r in fr)      # (ids of never-confirmed tracks are not in the output)
    # tracks returned by the fused device trackers are views: their state only changes inside the tracker's step
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    from yolov7_tracker_amd import _lib
    BaseTrack._count = 0
    bt = ByteTrack(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280, iou_thresh=0.5))
    view = bt.update(dets[0], None)[0]
    with pytest.raises(_lib.Y7TError):
        view.predict()
    assert bt.removed_stracks == [] and view.state == TrackState.Tracked


def test_empty_and_ragged_frames():
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    BaseTrack._count = 0
    t = ByteTrack(make_opts(), frame_rate=30)
    assert t.update(np.zeros((0, 6), np.float32), None) == []
    assert t.update_without_detection(None, None) == []
    one = np.array([[10, 10, 50, 90, 0.9, 3]], np.float32)
    assert t.update(one, None) == []
    cur = t.update(one, None)
    assert [c.track_id for c in cur] == [1] and cur[0].cls == 3
    assert t.update(np.zeros((0, 6), np.float32), None) == []


def test_capacity_overflow_is_loud():
    from yolov7_tracker_amd import _lib, synth
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    t = ByteTrack(make_opts(max_tracks=16, max_dets=128), frame_rate=30)
    with pytest.raises(_lib.Y7TError):
        for d in synth.make_detections(3, 80, seq_idx=0):
            t.update(d, None)


# ---------------------------------------------------------------- per-op kernels through the C ABI
def test_iou_cost_bit_exact_vs_oracle():
    from oracle import cnative
    from yolov7_tracker_amd.tracker import matching
    g = np.load(util.GOLDEN + "/lap_iou.npz")
    for k in range(int(g["n_cases"])):
        a, b = g["a%d" % k], g["b%d" % k]
        np.testing.assert_array_equal(matching.iou_distance(list(a), list(b)), 1 - g["iou%d" % k])
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 1280, (500, 4)); a[:, 2:] = a[:, :2] + rng.uniform(5, 200, (500, 2))
    b = np.round(a[rng.integers(0, 500, 500)] + rng.normal(0, 8, (500, 4)))
    np.testing.assert_array_equal(matching.iou_distance(list(a), list(b)), 1 - cnative.bbox_overlaps(a, b))
    assert matching.iou_distance([], list(b)).shape == (0, 500)


def test_lapjv_device_bit_exact_vs_oracle_and_golden():
    from oracle import cnative
    from yolov7_tracker_amd.tracker import matching
    g = np.load(util.GOLDEN + "/lap_iou.npz")
    for k in range(int(g["n_cases"])):
        cost = 1 - g["iou%d" % k]
        m, ua, ub = matching.linear_assignment(cost, float(g["lim%d" % k]))
        x = g["x%d" % k]
        want = np.asarray([[i, j] for i, j in enumerate(x) if j >= 0]).reshape(-1, 2)
        np.testing.assert_array_equal(np.asarray(m).reshape(-1, 2), want)
        np.testing.assert_array_equal(ua, np.where(x < 0)[0])
        np.testing.assert_array_equal(ub, np.where(g["y%d" % k] < 0)[0])
    rng = np.random.default_rng(2)
    for t in range(40):
        nr, nc = rng.integers(1, 70, 2)
        c = rng.random((nr, nc))
        if t % 2:
            c = np.where(rng.random((nr, nc)) < 0.8, 1.0, c)
        lim = [0.9, 0.5, 0.7][t % 3]
        opt0, x0, y0 = cnative.lapjv(c, extend_cost=True, cost_limit=lim)
        opt1, x1, y1 = matching.lapjv_device(c, lim)
        np.testing.assert_array_equal(x0, x1)
        np.testing.assert_array_equal(y0, y1)
        assert abs(opt0 - opt1) < 1e-12
        opt2, x2, y2 = matching.lapjv_host(c, lim)              # host-pointer entry: same kernel behind library-owned staging buffers
        np.testing.assert_array_equal(x0, x2)
        np.testing.assert_array_equal(y0, y2)
        assert abs(opt0 - opt2) < 1e-12
    m, ua, ub = matching.linear_assignment(np.zeros((0, 5)), 0.9)
    assert m.shape == (0, 2) and ua == () and ub == (0, 1, 2, 3, 4)


def test_lapjv_device_with_ties_follows_lapjv():
    """non-unique optima (a pair exactly at the limit, equal-cost alternatives of integer boxes): the device notices and re-solves with the literal
    lapjv, so x / y are the reference package's answer in those cases too"""
    from oracle import cnative
    from yolov7_tracker_amd.tracker import matching
    n_tied = 0
    for c, lim in util.tie_prone_iou_costs(np.random.default_rng(3), 200):
        _, x0, y0 = cnative.lapjv(c, extend_cost=True, cost_limit=lim)
        _, x1, y1 = matching.lapjv_device(c, lim)
        np.testing.assert_array_equal(x0, x1)
        np.testing.assert_array_equal(y0, y1)
        n_tied += int((c == lim).any())
    assert n_tied > 10


@pytest.mark.parametrize("seed,scene", util.sort_tie_scenes())
def test_device_sort_scenes_with_assignment_ties(seed, scene):
    """two scenes of the random-scene generator in which one frame's assignment has a tie (an IoU cost exactly at the limit; two detections at the
    same IoU from a fresh track): ids and boxes equal to the oracle's, which runs lap's algorithm"""
    import types
    from oracle import tracker_np
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack, BaseTracker
    dets = util.random_scene(seed, scene)
    BaseTrack._count = 0
    t = BaseTracker(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=640, iou_thresh=0.5))
    got = []
    for d in dets:
        cur = t.update_without_detection() if d is None else t.update(d, None)
        got.append([(c.track_id, c.tlwh, float(c.cls), float(c.score)) for c in cur])
    util.assert_same_tracks(got, tracker_np.run("sort", dets, kalman_format="default"), "seed %d scene %d" % (seed, scene))


@pytest.mark.parametrize("seed", [20, 62, 76, 3, 4])
def test_device_deepsort_equals_oracle_on_random_scenes(seed):
    """random DeepSORT scenes on the device against the oracle with the product's summation order pinned (see tests/test_hostsim.py,
    test_hostsim_deepsort_equals_oracle_on_random_scenes): seeds 20 and 62 hold rows with two candidate detections of exactly equal appearance cost
    (the step re-solves those with lapjv.cpp run literally), 76 is 512 wide, 3 and 4 are ordinary"""
    import types
    from oracle import tracker_np
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.deepsort import DeepSORT
    dets, fn, dim = util.random_deepsort_scene(seed)
    BaseTrack._count = 0
    t = DeepSORT(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=640, iou_thresh=0.5))
    t.get_feature = lambda tlbrs, img: fn(tlbrs)
    got = []
    for d in dets:
        cur = t.update_without_detection() if d is None else t.update(d, None)
        got.append([(c.track_id, c.tlwh, float(c.cls), float(c.score)) for c in cur])
    util.assert_same_tracks(got, tracker_np.run("deepsort", dets, feature_fn=fn, dot=tracker_np.dot_sequential), "seed %d (dim %d)" % (seed, dim))


def test_lapjv_device_500x500_optimal():
    """BASELINE config 3 size: properties instead of the O(n^3) oracle -- valid partial matching, every kept cost
    below the limit, total cost equal to scipy's optimum of the explicit extended problem."""
    from scipy.optimize import linear_sum_assignment
    from yolov7_tracker_amd.tracker import matching
    rng = np.random.default_rng(4)
    n = 500
    a = rng.uniform(0, 1280, (n, 4)); a[:, 2:] = a[:, :2] + rng.uniform(8, 120, (n, 2))
    b = np.round(a[rng.permutation(n)] + rng.normal(0, 5, (n, 4)))
    cost = matching.iou_distance(list(a), list(b))
    opt, x, y = matching.lapjv_device(cost, 0.9)
    sel = x >= 0
    assert len(set(x[sel])) == sel.sum()
    assert np.all(y[x[sel]] == np.where(sel)[0])
    assert np.all(cost[np.where(sel)[0], x[sel]] < 0.9)
    ext = np.full((2 * n, 2 * n), 0.45); ext[n:, n:] = 0; ext[:n, :n] = cost
    r, c = linear_sum_assignment(ext)
    tot = cost[np.where(sel)[0], x[sel]].sum() + 0.45 * ((x < 0).sum() + (y < 0).sum())
    assert abs(tot - ext[r, c].sum()) < 1e-8


@pytest.mark.parametrize("kind", ["default", "botsort", "strongsort"])
def test_kalman_kernels_match_reference_golden(kind):
    from yolov7_tracker_amd.tracker.basetrack import KALMAN_DICT
    kal = np.load(util.GOLDEN + "/kalman.npz")
    f = KALMAN_DICT[kind]()
    mean, cov, z, conf = (kal[kind + "_" + k] for k in ("mean", "cov", "z", "conf"))
    mp, cp = f.multi_predict(mean, cov)
    np.testing.assert_allclose(mp, kal[kind + "_pred_mean"], rtol=1e-13)
    np.testing.assert_allclose(cp, kal[kind + "_pred_cov"], rtol=1e-12, atol=1e-12)
    for i in range(len(mean)):
        c = conf[i] if kind == "strongsort" else 0.0
        pm, pc = f.project(mean[i], cov[i], c)
        np.testing.assert_allclose(pm, kal[kind + "_proj_mean"][i], rtol=1e-13)
        np.testing.assert_allclose(pc, kal[kind + "_proj_cov"][i], rtol=1e-12)
        um, uc = f.update(mean[i], cov[i], z[i], c)
        np.testing.assert_allclose(um, kal[kind + "_upd_mean"][i], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(uc, kal[kind + "_upd_cov"][i], rtol=1e-8, atol=1e-9)
        a, b = f.initiate(z[i].astype(np.float32))
        assert a.dtype == np.float32
        np.testing.assert_array_equal(a.astype(np.float64), kal[kind + "_init32_mean"][i])
        np.testing.assert_allclose(b, kal[kind + "_init32_cov"][i], rtol=1e-7)
        a, b = f.initiate(z[i])
        np.testing.assert_array_equal(a, kal[kind + "_init64_mean"][i])
        np.testing.assert_allclose(b, kal[kind + "_init64_cov"][i], rtol=1e-14)
        if kind != "botsort":
            np.testing.assert_allclose(f.gating_distance(mean[i], cov[i], z), kal[kind + "_gate4"][i], rtol=1e-9)
            np.testing.assert_allclose(f.gating_distance(mean[i], cov[i], z, True), kal[kind + "_gate2"][i], rtol=1e-9)


def test_multi_gmc_kernel_matches_numpy(L):
    """botsort.py:250-269 on 500 tracks through the C ABI"""
    from yolov7_tracker_amd import _lib
    rng = np.random.default_rng(12)
    n = 500
    mean = rng.normal(0, 50, (n, 8))
    A = rng.normal(0, 1, (n, 8, 8)); cov = np.einsum("nij,nkj->nik", A, A) + np.eye(8)
    H = np.array([[1.002, -0.004, 3.5], [0.003, 0.997, -1.25]])
    m, c, h = torch.from_numpy(mean).cuda(), torch.from_numpy(cov.reshape(n, 64)).cuda(), torch.from_numpy(H.reshape(6)).cuda()
    _lib.check(L.y7t_kf_multi_gmc_f64(_lib.ptr(m), _lib.ptr(c), _lib.ptr(h), n, _lib.stream_ptr()))
    R8 = np.kron(np.eye(4), H[:, :2])
    m0 = mean @ R8.T; m0[:, :2] += H[:, 2]
    c0 = R8 @ cov @ R8.T
    np.testing.assert_allclose(m.cpu().numpy(), m0, rtol=1e-13, atol=1e-12)
    np.testing.assert_allclose(c.cpu().numpy().reshape(n, 8, 8), c0, rtol=1e-12, atol=1e-11)


def test_kalman_batch_500_tracks_roundtrip_properties(L):
    """config-3 size: predict is linear in P (F P F^T + Q), update keeps P symmetric PSD and shrinks its trace."""
    from oracle.kalman_np import KalmanNP
    from yolov7_tracker_amd.tracker.basetrack import KALMAN_DICT
    rng = np.random.default_rng(8)
    n = 500
    f, o = KALMAN_DICT["default"](), KalmanNP("default")
    mean = np.zeros((n, 8)); mean[:, :2] = rng.uniform(0, 1280, (n, 2)); mean[:, 2] = rng.uniform(0.3, 2, n); mean[:, 3] = rng.uniform(10, 300, n)
    A = rng.normal(0, 1, (n, 8, 8)); cov = np.einsum("nij,nkj->nik", A, A) + np.eye(8)
    mp, cp = f.multi_predict(mean, cov)
    m0, c0 = o.multi_predict(mean, cov)
    np.testing.assert_allclose(mp, m0, rtol=1e-13)
    np.testing.assert_allclose(cp, c0, rtol=1e-12, atol=1e-12)
    assert np.abs(cp - cp.transpose(0, 2, 1)).max() < 1e-9


@pytest.mark.parametrize("kind,fmt", [("sort", "default"), ("bytetrack", "default"), ("botsort", "botsort")])
def test_fused_tracker_equals_oracle_on_random_scenes(kind, fmt):
    """seeded scenes of random density, miss / clutter rates, frame gaps, empty frames and camera warps: the device step and the
    numpy oracle (pinned to the reference) agree on every id in every frame"""
    from oracle import tracker_np
    from yolov7_tracker_amd import synth
    import zlib
    rng = np.random.default_rng(zlib.crc32(("%s/%s" % (kind, fmt)).encode()))      # (hash() of a str changes from process to process)
    for scene in range(6):
        n_obj = int(rng.integers(5, 150))
        n_frames = int(rng.integers(15, 40))
        dets = synth.make_detections(n_frames, n_obj, 640, seq_idx=200 + scene, miss=float(rng.uniform(0.0, 0.3)), fp=float(rng.uniform(0.0, 0.2)))
        gap = int(rng.integers(0, 9))
        if gap > 2:
            dets = [None if (i % gap == gap - 1) else d for i, d in enumerate(dets)]
        if scene % 3 == 2:
            dets[n_frames // 2] = np.zeros((0, 6), np.float32)
        warps = synth.make_warps(n_frames, seq_idx=scene) if kind == "botsort" else None
        want = tracker_np.run(kind, dets, kalman_format=fmt, warps=warps)
        got, _ = run_device_tracker(kind, fmt, dets, warps=warps)
        util.assert_same_tracks(got, want, "%s scene %d (%d objects, %d frames, gap %d)" % (kind, scene, n_obj, n_frames, gap))


def test_deepsort_empty_and_low_confidence_first_frames_then_real_embeddings():
    """frames without a detection above det_thresh before the first real one must not pin the feature width to a guess --
    real footage opens like this, and OSNet / the DeepSORT net are 512 wide.  Tracks only exist after a kept detection, so the feature state is
    re-made for the real width; once vectors are stored a change of width is an error."""
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.deepsort import DeepSORT
    BaseTrack._count = 0
    rng = np.random.default_rng(5)
    dim = [512]
    t = DeepSORT(make_opts(), frame_rate=30, reid_model=lambda crops: rng.normal(size=(len(crops), dim[0])).astype(np.float32))
    img = np.zeros((200, 300, 3), np.uint8)
    assert t.update(np.zeros((0, 6), np.float32), img) == []
    assert t.update(np.array([[10, 10, 50, 90, 0.1, 0]], np.float32), img) == []          # below det_thresh = 0.2: no features asked for
    one = np.array([[10, 10, 50, 90, 0.9, 3]], np.float32)
    t.update(one, img)
    assert t._feat_dim == 512
    t.update(one, img)
    cur = t.update(one, img)
    assert [c.track_id for c in cur] == [1] and cur[0].cls == 3
    dim[0] = 128
    with pytest.raises(ValueError):
        t.update(one, img)


def test_plain_step_refuses_a_deepsort_pool():
    """y7t_tracker_step on a pool initialised as DeepSORT would create / update tracks without the appearance rings (a reused
    slot keeps its previous occupant's vectors): frames with detections are refused with Y7T_E_STATE; the predict-only step is shared"""
    from yolov7_tracker_amd import _lib
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack, BaseTracker
    from yolov7_tracker_amd.tracker.deepsort import DeepSORT
    BaseTrack._count = 0
    t = DeepSORT(make_opts(), frame_rate=30, reid_model=lambda crops: np.ones((len(crops), 64), np.float32))
    with pytest.raises(_lib.Y7TError):
        BaseTracker._launch(t, np.array([[10, 10, 50, 90, 0.9, 3]], np.float32))
    assert t.update_without_detection(None, None) == []
    # the feature tensor of the device chain is validated too (a short / half tensor would be read out of bounds)
    with pytest.raises(_lib.Y7TError):
        t._launch(torch.zeros((2, 6), device="cuda"), torch.zeros((1, 64), device="cuda"))


def test_nms_max_det_beyond_the_lds_budget_is_an_argument_error():
    """k_nms_keep keeps max_det boxes in LDS; a huge max_det is refused as an argument error, not a bare launch failure"""
    import ctypes
    from yolov7_tracker_amd import _lib
    L = _lib.load()
    B, cap = 1, 256
    ws = torch.zeros(int(L.y7t_det_postprocess_workspace_bytes(B, cap, 30000)), dtype=torch.uint8, device="cuda")
    lb = torch.tensor([[1.0, 0, 0, 64, 64]], device="cuda")
    dets, nd, keep = torch.zeros((B, 4000, 6), device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros((B, 4000), dtype=torch.int32, device="cuda")
    rc = L.y7t_det_postprocess(None, None, None, None, None, 4, 3, 15, B, ctypes.c_float(0.01), ctypes.c_float(0.45), 4000, 30000, cap, _lib.ptr(lb),
                               _lib.ptr(dets), _lib.ptr(nd), _lib.ptr(keep), None, _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    assert rc == -1 and b"max_det" in L.y7t_last_error()


@pytest.mark.parametrize("kind,fmt", [("bytetrack", "default"), ("botsort", "botsort")])
def test_frames_in_one_launch_equal_frame_by_frame(kind, fmt):
    """y7t_tracker_step_frames: a batch of consecutive frames of one tracker in ONE launch gives exactly the rows that one launch per frame gives
    (empty frames and warps included); the oracle pins both"""
    from oracle import tracker_np
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.botsort import BoTSORT
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    n_frames, chunk = 36, 12
    dets = synth.make_detections(n_frames, 60, 640, seq_idx=9, miss=0.1, fp=0.1)
    dets[7] = np.zeros((0, 6), np.float32)
    warps = synth.make_warps(n_frames, seq_idx=9) if kind == "botsort" else None
    want = tracker_np.run(kind, dets, kalman_format=fmt, warps=warps)
    BaseTrack._count = 0
    t = (BoTSORT if kind == "botsort" else ByteTrack)(make_opts(kalman_format=fmt), frame_rate=30)
    dd = [torch.from_numpy(d).cuda() for d in dets]
    ww = [torch.from_numpy(np.ascontiguousarray(w, dtype=np.float64).reshape(6)).cuda() for w in warps] if warps is not None else None
    outs = torch.zeros((n_frames, t.cap_t + 1, 8), dtype=torch.float64, device="cuda")
    for f0 in range(0, n_frames, chunk):
        tab = t.frames_table(dd[f0:f0 + chunk], [outs[f] for f in range(f0, f0 + chunk)], ww[f0:f0 + chunk] if ww is not None else None)
        t._launch_frames(tab)
    torch.cuda.synchronize()
    h = outs.cpu().numpy()
    got = []
    for f in range(n_frames):
        c = int(h[f, t.cap_t].view(np.int32)[0])
        got.append([(int(r[0]), r[1:5].copy(), float(r[5]), float(r[6])) for r in h[f, :c]])
    util.assert_same_tracks(got, want, "%s, %d frames in launches of %d" % (kind, n_frames, chunk))
    assert t.frame_id == n_frames


@pytest.mark.parametrize("kind,extra", [("bytetrack", 0), ("bytetrack", 60), ("botsort", 0)])
def test_component_larger_than_a_wave_on_the_device(kind, extra):
    """the 160-track lattice of tests/util.lattice_scene: one connected component of 160 rows -- more than the 64 slots of the register-resident wave solve, so the
    same wave solves it with its state in the work arrays (y7t_assoc_sparse_try step 4a, the `ncl > 64 || nrw > 64` branch); ids and boxes of every frame equal to
    the oracle's (the host build of the same text: tests/test_hostsim.py).  Round 5's form of this path had never run on a device; round 6 ran it first under a
    timeout (scripts/debug_lattice.py, profiles/r06_large_components.txt)"""
    from oracle import tracker_np
    dets = util.lattice_scene(extra_cols=extra)
    fmt = "botsort" if kind == "botsort" else "default"
    want = tracker_np.run(kind, dets, kalman_format=fmt)
    got, _ = run_device_tracker(kind, fmt, dets, max_tracks=1024, max_dets=1024)
    util.assert_same_tracks(got, want, "lattice %s +%d" % (kind, extra))


@pytest.mark.parametrize("n_obj,size", [(250, 640), (400, 640), (400, 480)])
@pytest.mark.parametrize("kind", ["bytetrack", "botsort"])
def test_crowded_scenes_on_the_device(kind, n_obj, size):
    """NATURAL components larger than a wave: crowds of 250 / 400 objects on a 640 / 480 px frame, 10 % misses and clutter, camera warps for
    BoT-SORT -- the scenes of tests/test_hostsim.py::test_hostsim_crowded_scenes_through_every_solver_path, where the host build counts 1-2 of 42 associations (250
    objects) to a third of them (400) with a component of more than 64 rows or columns.  Every id and box of every frame equal to the oracle's, per frame launches
    and the frames of the scene in ONE launch (k_tracker_step_frames: the index lists in the LDS arena)"""
    from oracle import tracker_np
    from yolov7_tracker_amd import synth
    fmt = "botsort" if kind == "botsort" else "default"
    dets = synth.make_detections(14, n_obj, size, seq_idx=300 + n_obj, miss=0.1, fp=0.1)
    warps = synth.make_warps(14, seq_idx=3) if kind == "botsort" else None
    want = tracker_np.run(kind, dets, kalman_format=fmt, warps=warps)
    got, _ = run_device_tracker(kind, fmt, dets, warps=warps, max_tracks=2048, max_dets=1024)
    util.assert_same_tracks(got, want, "%s, %d objects on %d px" % (kind, n_obj, size))
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.botsort import BoTSORT
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    BaseTrack._count = 0
    t = (BoTSORT if kind == "botsort" else ByteTrack)(make_opts(kalman_format=fmt, max_tracks=2048, max_dets=1024), frame_rate=30)
    dd = [torch.from_numpy(d).cuda() for d in dets]
    ww = [torch.from_numpy(np.ascontiguousarray(w, dtype=np.float64).reshape(6)).cuda() for w in warps] if warps is not None else None
    outs = torch.zeros((len(dets), t.cap_t + 1, 8), dtype=torch.float64, device="cuda")
    t._launch_frames(t.frames_table(dd, [outs[f] for f in range(len(dets))], ww))
    torch.cuda.synchronize()
    h = outs.cpu().numpy()
    got = []
    for f in range(len(dets)):
        c = int(h[f, t.cap_t].view(np.int32)[0])
        got.append([(int(r[0]), r[1:5].copy(), float(r[5]), float(r[6])) for r in h[f, :c]])
    util.assert_same_tracks(got, want, "%s, %d objects on %d px, one launch" % (kind, n_obj, size))


def test_frames_launch_reads_the_row_counts_on_the_device():
    """the coupled hand-over (bench.py `coupled`): the rows of y7t_det_postprocess' (B, 300, 6) output as the frames' detections and its `ndets` array
    as the counts, both read by y7t_tracker_step_frames on the device -- here with padded rows of a scene whose frames have different counts (and garbage behind the
    count, which the step must not look at); ids and boxes equal to the oracle's"""
    from oracle import tracker_np
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    dets = synth.make_detections(24, 60, 640, seq_idx=4, miss=0.2, fp=0.1)
    dets[5] = np.zeros((0, 6), np.float32)
    want = tracker_np.run("bytetrack", dets)
    rows = torch.full((len(dets), 300, 6), 1e9, dtype=torch.float32)
    for f, d in enumerate(dets):
        rows[f, :len(d)] = torch.from_numpy(d)
    rows = rows.cuda()
    counts = torch.tensor([len(d) for d in dets], dtype=torch.int32).cuda()
    BaseTrack._count = 0
    t = ByteTrack(make_opts(), frame_rate=30)
    outs = torch.zeros((len(dets), t.cap_t + 1, 8), dtype=torch.float64, device="cuda")
    t._launch_frames(t.frames_table([rows[f] for f in range(len(dets))], [outs[f] for f in range(len(dets))], None, counts_dev=counts))
    torch.cuda.synchronize()
    assert t._status() == 0
    h = outs.cpu().numpy()
    got = []
    for f in range(len(dets)):
        c = int(h[f, t.cap_t].view(np.int32)[0])
        got.append([(int(r[0]), r[1:5].copy(), float(r[5]), float(r[6])) for r in h[f, :c]])
    util.assert_same_tracks(got, want, "frames launch with device-side row counts")
    with pytest.raises(Exception):
        t.frames_table([rows[0]], [outs[0]], None, counts_dev=counts.cpu())
