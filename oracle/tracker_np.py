"""oracle/tracker_np.py -- TEST INFRASTRUCTURE ONLY.

Compact numpy restatement of the reference's SORT and ByteTrack association state
machines, for use where /root/reference is absent (the GPU box) and as the "port"
CPU baseline in bench.py:

  * BaseTracker.update          /root/reference/tracker/basetrack.py:368-487   (kind='sort')
  * ByteTrack.update            /root/reference/tracker/bytetrack.py:41-204    (kind='bytetrack')
  * update_without_detection    /root/reference/tracker/basetrack.py:489-537
  * STrack activate/update/re_activate/tlwh/tlbr, multi_predict  basetrack.py:74-339
  * joint/sub/remove_duplicate_stracks                            basetrack.py:540-576
  * iou_distance / linear_assignment                              matching.py:30-82
  * DeepSORT.update, gate_cost_matrix, gated_metric               /root/reference/tracker/deepsort.py:43-224  (kind='deepsort')
    matching_cascade / nearest_embedding_distance / cal_cosine_distance   matching.py:105-127,165-178,216-277
    (the ReID network is replaced by a feature function at DeepSORT.get_feature, deepsort.py:19-41)

It is pinned against the reference's own modules (imported through
oracle/ref_harness.py) in tests/test_oracle_pinned.py and through tests/golden/.
The float32/float64 dtype flow of the reference (float32 `_tlwh`, float32 mean right
after `initiate`, float64 afterwards) is kept by using the same numpy dtypes.
"""
import numpy as np

from . import cnative
from .kalman_np import KalmanNP

NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3


class IdCounter:
    """Mirror of the process-global BaseTrack._count (basetrack.py:22,43-46)."""

    def __init__(self):
        self.count = 0

    def next(self):
        self.count += 1
        return self.count


GLOBAL_IDS = IdCounter()


class T:
    __slots__ = ("cls", "box", "score", "activated", "tid", "start", "frame", "tsu", "state", "mean", "cov", "fmt",
                 "len", "features")

    def __init__(self, cls, tlwh, score, fmt, feature=None):
        self.cls, self.score, self.fmt = cls, score, fmt
        self.features = [] if feature is None else [feature]        # basetrack.py:97-103
        self.box = np.asarray(tlwh, dtype=np.float32)
        self.activated, self.tid, self.start, self.frame, self.tsu = False, None, None, None, None
        self.state, self.mean, self.cov, self.len = NEW, None, None, 0

    @property
    def tlwh(self):
        if self.mean is None:
            return self.box.copy()
        r = self.mean[:4].copy()
        if self.fmt in ("default", "strongsort"):
            r[2] *= r[3]
            r[:2] -= r[2:] / 2
        elif self.fmt == "botsort":
            r[:2] -= r[2:] / 2
        else:
            raise NotImplementedError(self.fmt)
        return r

    @property
    def tlbr(self):
        r = self.tlwh.copy()
        r[2:] += r[:2]
        return r

    def meas(self, tlwh):
        r = np.asarray(tlwh).copy()
        if self.fmt in ("default", "strongsort"):
            r[:2] += r[2:] / 2
            r[2] /= r[-1]
        elif self.fmt == "botsort":
            r[:2] += r[2:] // 2
        else:
            raise NotImplementedError(self.fmt)
        return r


def _iou_dist(a, b):
    at = [t.tlbr for t in a]
    bt = [t.tlbr for t in b]
    if len(at) == 0 or len(bt) == 0:
        return np.zeros((len(at), len(bt)), dtype=np.float64)
    return 1 - cnative.bbox_overlaps(np.ascontiguousarray(at, dtype=np.float64),
                                     np.ascontiguousarray(bt, dtype=np.float64))


def _assign(cost, thresh):
    if cost.size == 0:
        return [], list(range(cost.shape[0])), list(range(cost.shape[1]))
    _, x, y = cnative.lapjv(cost, extend_cost=True, cost_limit=thresh)
    m = [(i, int(j)) for i, j in enumerate(x) if j >= 0]
    return m, [int(i) for i in np.where(x < 0)[0]], [int(j) for j in np.where(y < 0)[0]]


def _joint(a, b):
    seen, out = set(), []
    for t in a:
        seen.add(t.tid)
        out.append(t)
    for t in b:
        if t.tid not in seen:
            seen.add(t.tid)
            out.append(t)
    return out


def _sub(a, b):
    d = {}
    for t in a:
        d[t.tid] = t
    for t in b:
        if d.get(t.tid, 0):
            del d[t.tid]
    return list(d.values())


def _dedup(a, b):
    pd = _iou_dist(a, b)
    da, db = set(), set()
    for p, q in zip(*np.where(pd < 0.15)):
        if a[p].frame - a[p].start > b[q].frame - b[q].start:
            db.add(q)
        else:
            da.add(p)
    return [t for i, t in enumerate(a) if i not in da], [t for i, t in enumerate(b) if i not in db]


class TrackerNP:
    def __init__(self, kind="bytetrack", conf_thresh=0.2, track_buffer=30, kalman_format="default", iou_thresh=0.5,
                 frame_rate=30, ids=None):
        assert kind in ("sort", "bytetrack", "botsort", "deepsort")
        self.feature_fn = None            # deepsort: (N, 4) tlbr -> (N, D) appearance features (stands in for the ReID network)
        self.dot = np.dot                 # cal_cosine_distance's product (matching.py:178)
        self.kind, self.fmt = kind, kalman_format
        self.det_thresh = conf_thresh
        self.iou_thresh = iou_thresh
        self.max_time_lost = int(frame_rate / 30.0 * track_buffer)
        self.low_thresh = max(0.15, conf_thresh - 0.3)
        self.kf = KalmanNP(kalman_format)
        self.tracked, self.lost, self.removed = [], [], []
        self.frame_id = 0
        self.ids = ids if ids is not None else GLOBAL_IDS

    # --- per-track operations ------------------------------------------------
    def _activate(self, t):
        t.tid = self.ids.next()
        t.mean, t.cov = self.kf.initiate(t.meas(t.box), f32_std=True)
        t.state = TRACKED
        if self.frame_id == 1:
            t.activated = True
        t.frame = t.start = self.frame_id
        t.tsu = 0

    def _kf_update(self, t, det):
        z = t.meas(det.tlwh)
        if self.fmt == "strongsort":
            return self.kf.update(t.mean, t.cov, z, det.score)
        return self.kf.update(t.mean, t.cov, z)

    def _update(self, t, det):
        t.frame = self.frame_id
        t.len += 1
        t.score = det.score
        t.mean, t.cov = self._kf_update(t, det)
        if det.features:                                  # basetrack.py:324-332, use_avg_of_feature=False (deepsort.py:111)
            f = det.features[0] / np.linalg.norm(det.features[0])
            t.features.append(f)
            t.features = t.features[-100:]
        t.state, t.activated, t.tsu = TRACKED, True, 0

    def _reactivate(self, t, det):
        z = t.meas(det.tlwh)
        t.mean, t.cov = self.kf.update(t.mean, t.cov, z)  # no confidence: basetrack.py:283-285
        t.len = 0
        t.state, t.activated, t.frame, t.score, t.tsu = TRACKED, True, self.frame_id, det.score, 0

    def _multi_predict(self, pool):
        if pool:
            mm = np.asarray([t.mean.copy() for t in pool])
            cc = np.asarray([t.cov for t in pool])
            for i, t in enumerate(pool):
                if t.state != TRACKED:
                    mm[i][-1] = 0
            mm, cc = self.kf.multi_predict(mm, cc)
            for t, m, c in zip(pool, mm, cc):
                t.mean, t.cov = m, c
        for t in pool:
            t.tsu += 1

    def _mk(self, rows):
        return [T(r[5], np.array([r[0], r[1], r[2] - r[0], r[3] - r[1]], dtype=rows.dtype), r[4], self.fmt)
                for r in rows]

    def _finish(self, act, refind, lost, removed):
        self.tracked = [t for t in self.tracked if t.state == TRACKED]
        self.tracked = _joint(self.tracked, act)
        self.tracked = _joint(self.tracked, refind)
        self.lost = _sub(self.lost, self.tracked)
        self.lost.extend(lost)
        self.lost = _sub(self.lost, self.removed)
        self.removed.extend(removed)
        self.tracked, self.lost = _dedup(self.tracked, self.lost)
        return [t for t in self.tracked if t.activated]

    def _match_apply(self, tracks, dets, matches, act, refind, only_update=False):
        for it, idt in matches:
            t, d = tracks[it], dets[idt]
            if only_update or t.state == TRACKED:
                self._update(t, d)
                act.append(t)
            elif self.kind == "sort" or t.state == LOST:
                self._reactivate(t, d)
                refind.append(t)

    # --- frame step -----------------------------------------------------------
    @staticmethod
    def _multi_gmc(tracks, H):
        """botsort.py:250-269"""
        R = H[:2, :2]
        R8 = np.kron(np.eye(4, dtype=float), R)
        t = H[:2, 2]
        for tr in tracks:
            m = R8.dot(tr.mean.copy())
            m[:2] += t
            tr.mean, tr.cov = m, R8.dot(tr.cov).dot(R8.transpose())

    # --- DeepSORT (deepsort.py:43-224) ---------------------------------------------
    def _cos(self, a, b):
        """matching.py:165-178.  `self.dot` is np.dot like the reference; a test may set it to `dot_sequential` to pin the summation order
        (numpy's BLAS picks one by operand shape and CPU -- see that function)"""
        a = a / np.linalg.norm(a, axis=1, keepdims=True)
        b = b / np.linalg.norm(b, axis=1, keepdims=True)
        return self.dot(a, b.T)

    def _gated_metric(self, tracks, dets):
        """nearest_embedding_distance (matching.py:105-127) + gate_cost_matrix (deepsort.py:43-66)"""
        cost = np.zeros((len(tracks), len(dets)))
        df = np.asarray([d.features[-1] for d in dets])
        for r, t in enumerate(tracks):
            cost[r, :] = (1. - self._cos(np.asarray(t.features), df)).min(axis=0)
        zs = np.asarray([d.meas(d.tlwh) for d in dets])            # STrack.tlwh2xyah of every detection
        cost[cost > 0.15] = 1e5
        for r, t in enumerate(tracks):
            cost[r, self.kf.gating_distance(t.mean, t.cov, zs, False) > 9.4877] = 1e5      # chi2inv95[4]
        return cost

    def _cascade(self, tracks, dets):
        """matching.matching_cascade(gated_metric, 0.9, max_time_lost, tracks, dets) (matching.py:216-277)"""
        to_match = list(range(len(dets)))
        matches = []
        for level in range(self.max_time_lost):
            if not to_match:
                break
            tl = [k for k in range(len(tracks)) if tracks[k].tsu == 1 + level]
            if not tl:
                continue
            m, _, ucol = _assign(self._gated_metric([tracks[k] for k in tl], [dets[k] for k in to_match]), 0.9)
            matches += [(tl[r], to_match[c]) for r, c in m]
            to_match = [to_match[c] for c in ucol]
        um = list(set(range(len(tracks))) - set(k for k, _ in matches))
        return matches, um, to_match

    def _update_deepsort(self, det):
        act, refind, lost, removed = [], [], [], []
        rows = det[det[:, 4] > self.det_thresh]
        dets = []
        if len(rows):
            feats = self.feature_fn(rows[:, :4])
            dets = [T(r[5], np.array([r[0], r[1], r[2] - r[0], r[3] - r[1]], dtype=rows.dtype), r[4], self.fmt, feature=f)
                    for r, f in zip(rows, feats)]
        unconf = [t for t in self.tracked if not t.activated]
        conf = [t for t in self.tracked if t.activated]
        pool = _joint(conf, self.lost)
        self._multi_predict(pool)
        m, ut0, ud0 = self._cascade(pool, dets)
        self._match_apply(pool, dets, m, act, refind)
        tr0 = [pool[i] for i in ut0 if pool[i].state == TRACKED]
        d0 = [dets[i] for i in ud0]
        m, ut1, ud1 = _assign(_iou_dist(tr0, d0), 0.5)
        d1 = [d0[i] for i in ud1]
        self._match_apply(tr0, d0, m, act, refind)
        for i in ut1:
            t = pool[i]                                   # deepsort.py:171-173 indexes strack_pool with an index into u_tracks0 (sic)
            t.state = LOST
            lost.append(t)
        m, ut2, ud2 = _assign(_iou_dist(unconf, d1), 0.9)
        self._match_apply(unconf, d1, m, act, refind, only_update=True)
        for i in ut2:
            unconf[i].state = REMOVED
            removed.append(unconf[i])
        for i in ud2:
            if d1[i].score > self.det_thresh:
                self._activate(d1[i])
                act.append(d1[i])
        for t in self.lost:
            if self.frame_id - t.frame > self.max_time_lost:       # end_frame == frame_id of the last update
                t.state = REMOVED
                removed.append(t)
        return self._finish(act, refind, lost, removed)

    def update(self, det, warp=None):
        det = np.asarray(det, dtype=np.float32).reshape(-1, 6)
        self.frame_id += 1
        if self.kind == "deepsort":
            return self._update_deepsort(det)
        act, refind, lost, removed = [], [], [], []
        unconf = [t for t in self.tracked if not t.activated]
        conf = [t for t in self.tracked if t.activated]
        if self.kind == "sort":
            dets = self._mk(det[det[:, 4] > self.det_thresh])
            pool = _joint(conf, self.lost)
            self._multi_predict(pool)
            m, ut, ud = _assign(_iou_dist(pool, dets), self.iou_thresh)
            self._match_apply(pool, dets, m, act, refind)
            for i in ut:
                if pool[i].state == TRACKED:
                    pool[i].state = LOST
                    lost.append(pool[i])
            left = [dets[i] for i in ud]
            m, ut, ud = _assign(_iou_dist(unconf, left), self.iou_thresh + 0.1)
            self._match_apply(unconf, left, m, act, refind)
            gate = self.det_thresh + 0.1
        else:
            hi = det[:, 4] >= self.det_thresh
            lo = np.logical_and(np.logical_not(hi), det[:, 4] > self.low_thresh)
            d_hi, d_lo = self._mk(det[hi]), self._mk(det[lo])
            pool = _joint(conf, self.lost)
            self._multi_predict(pool)
            if self.kind == "botsort" and warp is not None:      # botsort.py:383-386
                H = np.asarray(warp, dtype=np.float64).reshape(2, 3)
                self._multi_gmc(pool, H)
                self._multi_gmc(unconf, H)
            m, ut, ud = _assign(_iou_dist(pool, d_hi), 0.9)
            self._match_apply(pool, d_hi, m, act, refind)
            # bytetrack.py:131 keeps only still-Tracked leftovers; botsort.py:411 keeps all of them
            rem = [pool[i] for i in ut if self.kind == "botsort" or pool[i].state == TRACKED]
            left = [d_hi[i] for i in ud]
            m, ut2, _ = _assign(_iou_dist(rem, d_lo), 0.5)
            self._match_apply(rem, d_lo, m, act, refind)
            for i in ut2:
                rem[i].state = LOST
                lost.append(rem[i])
            m, ut, ud = _assign(_iou_dist(unconf, left), 0.7)
            self._match_apply(unconf, left, m, act, refind, only_update=True)
            gate = self.det_thresh + 0.1
        for i in ut:
            unconf[i].state = REMOVED
            removed.append(unconf[i])
        for i in (range(len(left)) if self.kind == "botsort" else ud):     # botsort.py:462-466 iterates u_dets0_idx (sic)
            if left[i].score > gate:
                self._activate(left[i])
                act.append(left[i])
        for t in self.lost:
            if self.frame_id - t.frame > self.max_time_lost:
                t.state = REMOVED
                removed.append(t)
        return self._finish(act, refind, lost, removed)

    def update_without_detection(self):
        self.frame_id += 1
        conf = [t for t in self.tracked if t.activated]
        self._multi_predict(_joint(conf, self.lost))
        return self._finish([], [], [], [])


def dot_sequential(a, bt):
    """float32 a @ bt with every product ONE fused-multiply-add chain over k = 0 .. K-1 -- the order OpenBLAS's blocked sgemm kernel uses for
    operands of more than a few rows with K <= one panel, and the order the device kernels use for every shape.  np.dot itself switches order with
    the operand shapes (gemv / sdot for one-row operands, a small-matrix kernel, K panels beyond 384) and with the CPU the wheel dispatches on, so
    the last bit of a cosine distance is not a property of the reference; tests that need the cascade's near-ties (costs one ulp apart) decided
    the same way on both sides pin the order with this function.  (float64 product of two float32 is exact; the one rounding is the fma's)"""
    acc = np.zeros((a.shape[0], bt.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc = (a[:, k:k + 1].astype(np.float64) * bt[k][None, :].astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    return acc


def run(kind, dets_per_frame, ids=None, warps=None, feature_fn=None, dot=None, **kw):
    """-> per-frame list of (track_id, tlwh float64[4], cls, score), like ref_harness.run_reference_tracker."""
    trk = TrackerNP(kind, ids=ids if ids is not None else IdCounter(), **kw)
    trk.feature_fn = feature_fn
    if dot is not None:
        trk.dot = dot
    out = []
    for fi, det in enumerate(dets_per_frame):
        cur = trk.update_without_detection() if det is None else trk.update(det, None if warps is None else warps[fi])
        out.append([(int(t.tid), np.asarray(t.tlwh, dtype=np.float64).copy(), float(t.cls), float(t.score))
                    for t in cur])
    return out
