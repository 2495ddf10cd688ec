"""Tracker plugin surface of the reference (/root/reference/tracker/basetrack.py): TrackState,
BaseTrack (process-global id counter :21-61), STrack (:74-339) and BaseTracker == SORT (:346-537),
re-designed for the MI355X: the track pool (Kalman means/covariances, states, the
tracked/lost lists) lives in ONE device-resident struct-of-arrays blob and a whole
`update()` -- multi_predict, the IoU cost matrices, the linear assignments, the Kalman updates
and the list bookkeeping -- is one kernel launch of liby7t.so (y7t_tracker_step).  The Python
objects here are views: `update()` returns STrack views built from the rows the kernel wrote;
everything else (`mean`, `cov`, `state`, `tracked_stracks`, ...) is read back lazily.
"""
import ctypes
import weakref

import numpy as np
import torch

from .. import _lib
from . import matching  # noqa: F401  (re-exported like the reference module does)
from .kalman_filter import KalmanFilter, NaiveKalmanFilter, BoTSORTKalmanFilter, NSAKalmanFilter


class TrackState(object):
    New = 0
    Tracked = 1
    Lost = 2
    Removed = 3


class _IdCounter:
    """BaseTrack._count on the device: one int32 shared by every tracker of the process."""
    _tensor = None

    @classmethod
    def tensor(cls):
        if cls._tensor is None:
            _lib.require_gpu()
            cls._tensor = torch.zeros(1, dtype=torch.int32, device="cuda")
        return cls._tensor

    @classmethod
    def get(cls):
        return int(cls.tensor().item()) if cls._tensor is not None or torch.cuda.is_available() else 0

    @classmethod
    def set(cls, v):
        cls.tensor().fill_(int(v))


class _BaseTrackMeta(type):
    @property
    def _count(cls):
        return _IdCounter.get()

    @_count.setter
    def _count(cls, v):
        _IdCounter.set(v)


class BaseTrack(object, metaclass=_BaseTrackMeta):
    track_id = 0
    is_activated = False
    state = TrackState.New
    score = 0
    start_frame = 0
    frame_id = 0
    time_since_update = 0
    location = (np.inf, np.inf)

    @property
    def end_frame(self):
        return self.frame_id

    @staticmethod
    def next_id():
        BaseTrack._count = BaseTrack._count + 1
        return BaseTrack._count

    def activate(self, *args):
        raise NotImplementedError

    def predict(self):
        raise NotImplementedError

    def update(self, *args, **kwargs):
        raise NotImplementedError

    def mark_lost(self):
        self.state = TrackState.Lost

    def mark_removed(self):
        self.state = TrackState.Removed


KALMAN_DICT = {
    'default': KalmanFilter,
    'naive': NaiveKalmanFilter,
    'botsort': BoTSORTKalmanFilter,
    'strongsort': NSAKalmanFilter,
}
_KIND = {'default': 0, 'naive': 1, 'botsort': 2, 'strongsort': 3}


class STrack(BaseTrack):
    """Either a detection (constructed like the reference: STrack(cls, tlwh, score, kalman_format))
    or a view of one slot of a device track pool (built by the tracker)."""

    def __init__(self, cls, tlwh, score, kalman_format='default', feature=None, use_avg_of_feature=True,
                 store_features_budget=100):
        self.cls = cls
        self._tlwh = np.asarray(tlwh, dtype=np.float32)
        self.score = score
        self.is_activated = False
        self.tracklet_len = 0
        self.track_id = None
        self.start_frame = None
        self.frame_id = None
        self.time_since_update = None
        self.features = [] if feature is None else [feature]
        self.store_features_budget = store_features_budget
        self.has_feature = feature is not None
        self.use_avg_of_feature = use_avg_of_feature
        self.kalman_format = kalman_format
        self._kalman = None          # KALMAN_DICT[kalman_format](), created on first use (device-backed)
        self.mean, self.cov = None, None
        self._pool = None

    # -- the per-track plugin methods of the reference (basetrack.py:222-339): the same state transitions on host-side objects, the
    # Kalman arithmetic in liby7t.so's per-op kernels (tracker/kalman_filter.py).  The fused device trackers of this package do not go
    # through them (one kernel launch per frame); they exist so that a tracker written against the reference's STrack contract runs. ----
    @property
    def kalman(self):
        if self._kalman is None:
            self._kalman = KALMAN_DICT[self.kalman_format]()
        return self._kalman

    def _measurement(self, tlwh):
        if self.kalman_format in ('default', 'strongsort'):
            return self.tlwh2xyah(tlwh)
        if self.kalman_format == 'naive':
            return self.tlwh2xyar(tlwh)
        if self.kalman_format == 'botsort':
            return self.tlwh2xywh(tlwh)
        raise ValueError(self.kalman_format)

    def activate(self, frame_id):
        """basetrack.py:222-245"""
        self.track_id = BaseTrack.next_id()
        self.mean, self.cov = self.kalman.initiate(self._measurement(self._tlwh))
        self.state = TrackState.Tracked
        if frame_id == 1:
            self.is_activated = True
        self.frame_id = frame_id
        self.start_frame = frame_id
        self.time_since_update = 0

    def predict(self):
        """basetrack.py:247-251"""
        self.mean, self.cov = self.kalman.predict(self.mean, self.cov)

    @staticmethod
    def multi_predict(stracks, kalman):
        """basetrack.py:253-271: zero the last state component of non-Tracked tracks, one batched predict"""
        if len(stracks) > 0:
            multi_mean = np.asarray([st.mean.copy() for st in stracks])
            multi_covariance = np.asarray([st.cov for st in stracks])
            for i, st in enumerate(stracks):
                if st.state != TrackState.Tracked:
                    multi_mean[i][-1] = 0
            multi_mean, multi_covariance = kalman.multi_predict(multi_mean, multi_covariance)
            for i, (mean, cov) in enumerate(zip(multi_mean, multi_covariance)):
                stracks[i].mean = mean
                stracks[i].cov = cov
        for strack in stracks:
            strack.time_since_update += 1

    def re_activate(self, new_track, frame_id, new_id=False):
        """basetrack.py:273-294"""
        self.mean, self.cov = self.kalman.update(self.mean, self.cov, self._measurement(new_track.tlwh))
        self.tracklet_len = 0
        self.state = TrackState.Tracked
        self.is_activated = True
        self.frame_id = frame_id
        if new_id:
            self.track_id = self.next_id()
        self.score = new_track.score
        self.time_since_update = 0

    def update(self, new_track, frame_id):
        """basetrack.py:296-339"""
        self.frame_id = frame_id
        self.tracklet_len += 1
        new_tlwh = new_track.tlwh
        self.score = new_track.score
        measurement = self._measurement(new_tlwh)
        if self.kalman_format == 'strongsort':
            self.mean, self.cov = self.kalman.update(self.mean, self.cov, measurement, self.score)
        else:
            self.mean, self.cov = self.kalman.update(self.mean, self.cov, measurement)
        if new_track.has_feature:
            feature = new_track.features[0] / np.linalg.norm(new_track.features[0])
            if self.use_avg_of_feature:
                smooth_feat = 0.9 * self.features[-1] + (1 - 0.9) * feature
                smooth_feat /= np.linalg.norm(smooth_feat)
                self.features = [smooth_feat]
            else:
                self.features.append(feature)
                self.features = self.features[-self.store_features_budget:]
        self.state = TrackState.Tracked
        self.is_activated = True
        self.time_since_update = 0

    # -- converters (basetrack.py:110-181) -----------------------------------------------------
    @staticmethod
    def tlbr2tlwh(tlbr):
        r = np.asarray(tlbr).copy()
        r[2] -= r[0]
        r[3] -= r[1]
        return r

    @staticmethod
    def tlwh2xyah(tlwh):
        r = np.asarray(tlwh).copy()
        r[:2] += r[2:] / 2
        r[2] /= r[-1]
        return r

    @staticmethod
    def tlwh2xyar(tlwh):
        r = np.asarray(tlwh).copy()
        r[:2] += r[2:] / 2
        r[2] *= r[3]
        r[3] = tlwh[-1] / tlwh[-2]
        return r

    @staticmethod
    def tlwh2xywh(tlwh):
        r = np.asarray(tlwh).copy()
        r[:2] += r[2:] // 2
        return r

    @property
    def tlwh(self):
        if self.mean is None:
            return self._tlwh.copy()
        r = np.asarray(self.mean[:4]).copy()
        if self.kalman_format in ('default', 'strongsort'):
            r[2] *= r[3]
        r[:2] -= r[2:] / 2
        return r

    @property
    def tlbr(self):
        r = self.tlwh.copy()
        r[2:] += r[:2]
        return r

    def __repr__(self):
        return 'OT_{}_({}-{})'.format(self.track_id, self.start_frame, self.end_frame)


class _PoolTrack(STrack):
    """View of one slot of a tracker's device pool.  `track_id`, `tlwh`, `cls`, `score` come from the rows
    the step kernel returned; every other attribute is read back lazily from a snapshot of the pool."""

    def __init__(self, tracker, slot, track_id, tlwh, kcls, score):  # noqa: super().__init__ deliberately not called
        self._pool, self._slot, self._epoch = tracker, int(slot), tracker.frame_id
        self.track_id, self._tlwh_now, self.cls, self.score = int(track_id), np.asarray(tlwh, dtype=np.float64), kcls, score
        self.kalman_format = tracker.opts.kalman_format
        self.features, self.has_feature = [], False

    def _snap(self, name):
        if self._pool._snapshot()["tid"][self._slot] != self.track_id:
            raise _lib.Y7TError("track %d is no longer in the device pool" % self.track_id)
        return self._pool._snapshot()[name][self._slot]

    mean = property(lambda self: self._snap("mean").copy())
    cov = property(lambda self: self._snap("cov").reshape(8, 8).copy())
    state = property(lambda self: int(self._snap("state")))
    is_activated = property(lambda self: bool(self._snap("act")))
    frame_id = property(lambda self: int(self._snap("frame")))
    start_frame = property(lambda self: int(self._snap("start")))
    time_since_update = property(lambda self: int(self._snap("tsu")))
    tracklet_len = property(lambda self: int(self._snap("len")))
    _tlwh = property(lambda self: self._snap("box").copy())

    @property
    def tlwh(self):
        if self._epoch == self._pool.frame_id:
            return self._tlwh_now.copy()
        return STrack.tlwh.fget(self)

    def _view_only(self, *a, **k):
        raise _lib.Y7TError("this track is a view of a device track pool: its state changes only inside the tracker's fused step "
                            "(y7t_tracker_step); detached STrack objects support activate / predict / update / re_activate")
    activate = predict = update = re_activate = mark_lost = mark_removed = _view_only


class BaseTracker(object):
    """SORT (basetrack.py:346-537).  opts: conf_thresh, track_buffer, kalman_format, img_size, iou_thresh
    (+ optional max_tracks / max_dets capacities of the device pool)."""
    _KIND = 0  # Y7T_TRACKER_SORT

    def __init__(self, opts, frame_rate=30, *args, **kwargs):
        _lib.require_gpu()
        self._L = _lib.load()
        self.opts = opts
        self.frame_id = 0
        self.det_thresh = opts.conf_thresh
        self.buffer_size = int(frame_rate / 30.0 * opts.track_buffer)
        self.max_time_lost = self.buffer_size
        self.NMS = True
        if opts.kalman_format not in KALMAN_DICT or opts.kalman_format == 'naive':
            raise NotImplementedError("kalman_format %r is not implemented on the device" % (opts.kalman_format,))
        self.kalman = KALMAN_DICT[opts.kalman_format]()
        if isinstance(opts.img_size, int):
            self.model_img_size = [opts.img_size, opts.img_size]
        elif isinstance(opts.img_size, (list, tuple)):
            self.model_img_size = opts.img_size
        self.debug_mode = False
        self.cap_t = int(getattr(opts, "max_tracks", 1024))
        self.cap_d = int(getattr(opts, "max_dets", 1024))
        self.threads = int(getattr(opts, "tracker_threads", 0))
        nbytes = int(self._L.y7t_tracker_state_bytes(self.cap_t, self.cap_d))
        self._state = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        # the release is tied to the BLOB's lifetime, not to this object's: a copied tracker object shares the tensor, and dropping one of the two must not take the
        # kind / arena notes away from the survivor
        weakref.finalize(self._state, BaseTracker._release_state, self._L, int(self._state.data_ptr()))
        # numpy >= 2 keeps the float32 dtype of a freshly initiated mean; follow the
        # reference as it runs in this environment
        self._flags = 1 if int(np.__version__.split(".")[0]) >= 2 else 0
        _lib.check(self._L.y7t_tracker_init(_lib.ptr(self._state), nbytes, self._KIND, _KIND[opts.kalman_format], self.cap_t,
                                            self.cap_d, float(opts.conf_thresh), float(getattr(opts, "iou_thresh", 0.5)),
                                            self.max_time_lost, self._flags, _lib.ptr(_IdCounter.tensor()),
                                            _lib.stream_ptr()))
        # rows 0..cap_t-1: returned tracks; the int at the start of row cap_t: their count
        self._out = torch.zeros((self.cap_t + 1, 8), dtype=torch.float64, device="cuda")
        self._out_host = torch.zeros((self.cap_t + 1, 8), dtype=torch.float64).pin_memory()
        self._count_ptr = ctypes.c_void_p(self._out.data_ptr() + self.cap_t * 8 * 8)
        n = self._L.y7t_tracker_layout(self.cap_t, self.cap_d, None, 0)
        offs = (ctypes.c_int64 * n)()
        self._L.y7t_tracker_layout(self.cap_t, self.cap_d, offs, n)
        self._layout = {self._L.y7t_tracker_field_name(i).decode(): int(offs[i]) for i in range(n)}
        self._snap_cache = None
        self._det_keep = None
        # the frame-by-frame path (update(): one device round trip per frame, the reference's Timer semantics): detections go up through a pinned staging
        # buffer and the pool's status word comes down beside the rows -- ONE synchronisation per frame instead of a pageable copy + two round trips
        self._det_host = torch.zeros((self.cap_d, 6), dtype=torch.float32).pin_memory()
        self._det_dev = torch.zeros((self.cap_d, 6), dtype=torch.float32, device="cuda")
        self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._up_stream = self._up_event = None

    # ------------------------------------------------------------------------------------------

    @staticmethod
    def _release_state(L, address):
        # the state blob has returned to the allocator: the library drops its host-side notes for that address (y7t_tracker_release)
        try:
            L.y7t_tracker_release(ctypes.c_void_p(address))
        except Exception:
            pass

    def _launch(self, det_results, out=None, n_dev=None, warp=None, staged=False):
        """enqueue one frame step (asynchronous).  out: optional (cap_t + 1, 8) float64 device tensor that receives the
        returned rows (row cap_t holds the count) instead of the tracker's own buffer -- lets a pipeline keep every
        frame's result on the device without a per-frame host round trip."""
        if det_results is None:
            n, dptr = -1, None
        else:
            if isinstance(det_results, torch.Tensor):
                d = det_results.detach()
                if d.device.type != "cuda" or d.dtype != torch.float32 or not d.is_contiguous():
                    d = d.to(device="cuda", dtype=torch.float32).contiguous()
            elif staged:      # (the caller synchronises before the next frame: the staging buffers are free again by then)
                a = np.asarray(det_results, dtype=np.float32).reshape(-1, 6)
                if a.shape[0] > self.cap_d:
                    raise _lib.Y7TError("%d detections exceed the pool capacity max_dets=%d" % (a.shape[0], self.cap_d))
                self._det_host[:a.shape[0]].numpy()[...] = a
                d = self._det_dev[:a.shape[0]]
                # the rows go up on a copy stream of their own and the step waits for that copy's event: beside a detector forward already enqueued on the caller's
                # stream (track.py's loop, bench.py's latency mode) the upload overlaps it instead of sitting between the forward's last kernel and the step
                cur = torch.cuda.current_stream()
                if self._up_stream is None:
                    self._up_stream, self._up_event = torch.cuda.Stream(), torch.cuda.Event()
                with torch.cuda.stream(self._up_stream):
                    d.copy_(self._det_host[:a.shape[0]], non_blocking=True)
                    self._up_event.record(self._up_stream)
                cur.wait_event(self._up_event)
            else:
                d = torch.as_tensor(np.ascontiguousarray(det_results, dtype=np.float32)).cuda()
            d = d.reshape(-1, 6)
            self._det_keep = d  # keep the buffer alive until the kernel has consumed it
            n, dptr = d.shape[0], _lib.ptr(d)
            if n > self.cap_d:
                raise _lib.Y7TError("%d detections exceed the pool capacity max_dets=%d" % (n, self.cap_d))
        if out is None:
            optr, cptr = _lib.ptr(self._out), self._count_ptr
        else:
            optr, cptr = _lib.ptr(out), ctypes.c_void_p(out.data_ptr() + self.cap_t * 8 * 8)
        _lib.check(self._L.y7t_tracker_step(_lib.ptr(self._state), dptr, n, optr, self.cap_t, cptr, self.threads, _lib.ptr(warp),
                                            _lib.stream_ptr()))
        self.frame_id += 1
        self._snap_cache = None

    def frames_table(self, dets_dev, outs, warps=None, counts_dev=None):
        """device pointer tables of y7t_tracker_step_frames for a fixed set of consecutive frames (build once, launch many times):
        dets_dev = list of (n, 6) float32 device tensors, outs = list of (cap_t + 1, 8) float64 device tensors (row cap_t receives the count),
        warps = optional list of 2x3 float64 device tensors.  counts_dev: an int32 device tensor with one row count per frame that the launch reads ON THE
        DEVICE (e.g. `ndets` of Detector.postprocess, with dets_dev = the rows of its `dets`): the detector's NMS output feeds the tracker without a host
        round trip; each dets_dev[i] then only gives the address and the capacity of frame i's rows"""
        ds = [d.reshape(-1, 6) for d in dets_dev]
        for d in ds:
            if d.device.type != "cuda" or d.dtype != torch.float32 or not d.is_contiguous():
                raise _lib.Y7TError("frames_table: detections must be contiguous float32 device tensors")
            if d.shape[0] > self.cap_d:
                raise _lib.Y7TError("%d detections exceed the pool capacity max_dets=%d" % (d.shape[0], self.cap_d))
        n = len(ds)
        tab = torch.tensor([[d.data_ptr() for d in ds], [o.data_ptr() for o in outs], [o.data_ptr() + self.cap_t * 8 * 8 for o in outs],
                            [w.data_ptr() for w in warps] if warps is not None else [0] * n], dtype=torch.int64).cuda()
        if counts_dev is not None:
            if counts_dev.device.type != "cuda" or counts_dev.dtype != torch.int32 or not counts_dev.is_contiguous() or counts_dev.numel() < n:
                raise _lib.Y7TError("frames_table: counts_dev must be a contiguous int32 device tensor with one entry per frame")
            cnt = counts_dev
        else:
            cnt = torch.tensor([d.shape[0] for d in ds], dtype=torch.int32).cuda()
        nmax = max([d.shape[0] for d in ds] + [0])
        threads = self.threads if self.threads else (512 if nmax > 384 else 256)      # the library's rule for one frame (csrc/y7t_tracker.hip::step_threads)
        return (tab, cnt, n, warps is not None, (ds, outs, warps), threads)      # (the buffers stay alive with the table)

    def _launch_frames(self, table):
        """enqueue the frame steps of several CONSECUTIVE frames as ONE launch (y7t_tracker_step_frames; `table` from frames_table): exactly what `_launch`
        frame by frame produces -- for pipelines that hold a batch's detections before the tracker runs (bench.py, track.py --batch)"""
        tab, cnt, n, has_warps, _, threads = table
        _lib.check(self._L.y7t_tracker_step_frames(_lib.ptr(self._state), _lib.ptr(tab[0]), _lib.ptr(cnt), _lib.ptr(tab[1]), _lib.ptr(tab[2]), self.cap_t, n,
                                                   threads, _lib.ptr(tab[3]) if has_warps else None, _lib.stream_ptr()))
        self.frame_id += n
        self._snap_cache = None

    def _collect(self):
        off = self._layout["hdr_status"]
        self._out_host.copy_(self._out, non_blocking=True)
        self._status_host.copy_(self._state[off:off + 4].view(torch.int32), non_blocking=True)
        torch.cuda.current_stream().synchronize()
        host = self._out_host.numpy()
        cnt = int(host[self.cap_t].view(np.int32)[0])
        st = int(self._status_host[0])
        if st:
            raise _lib.Y7TError("device track pool overflow (status %d): raise opts.max_tracks / opts.max_dets" % st)
        # the frame's track objects.  Built column-wise (round 6): one conversion per COLUMN and a dict literal per object instead of five numpy scalar
        # conversions and ten attribute stores per row -- the per-row form cost 185 us of host time per 76-track frame, 7 % of the batch-1 frame the
        # reference's Timer brackets (bench.py: latency_mode).  Same attributes, same types as _PoolTrack.__init__ (cls / score: numpy float32 scalars;
        # tlwh: float64, a row of this call's OWN copy -- the pinned buffer is overwritten by the next frame).
        rows = host[:cnt]
        ids, slots = rows[:, 0].astype(np.int64).tolist(), rows[:, 7].astype(np.int64).tolist()
        boxes, kcls, score = list(rows[:, 1:5].copy()), list(rows[:, 5].astype(np.float32)), list(rows[:, 6].astype(np.float32))
        fid, kf, new = self.frame_id, self.opts.kalman_format, object.__new__
        out = []
        for slot, tid, box, c, sc in zip(slots, ids, boxes, kcls, score):
            o = new(_PoolTrack)
            o.__dict__ = {"_pool": self, "_slot": slot, "_epoch": fid, "track_id": tid, "_tlwh_now": box, "cls": c, "score": sc, "kalman_format": kf,
                          "features": [], "has_feature": False}
            out.append(o)
        return out

    def _status(self):
        off = self._layout["hdr_status"]
        return int(self._state[off:off + 4].view(torch.int32).item())

    def update(self, det_results, ori_img=None):
        """(N,6) [x1,y1,x2,y2,conf,cls] tensor/ndarray -> list of tracks (basetrack.py:368-487)."""
        self._launch(det_results, staged=True)
        return self._collect()

    def update_without_detection(self, det_results=None, ori_img=None):
        """basetrack.py:489-537: predict only."""
        self._launch(None)
        return self._collect()

    # -- lazy host views -----------------------------------------------------------------------
    def _snapshot(self):
        if self._snap_cache is None:
            Lo = self._layout
            end = Lo["lost"] + 4 * self.cap_t
            raw = self._state[:end].cpu().numpy()
            T = self.cap_t

            def arr(name, dtype, width):
                return raw[Lo[name]:Lo[name] + T * width * np.dtype(dtype).itemsize].view(dtype).reshape(T, width) if width > 1 \
                    else raw[Lo[name]:Lo[name] + T * np.dtype(dtype).itemsize].view(dtype)
            s = {"mean": arr("mean", np.float64, 8), "cov": arr("cov", np.float64, 64), "box": arr("box", np.float32, 4),
                 "score": arr("score", np.float32, 1), "cls": arr("cls", np.float32, 1)}
            for k in ("tid", "start", "frame", "tsu", "state", "act", "len", "inrem", "tracked", "lost"):
                s[k] = arr(k, np.int32, 1)
            for k in ("hdr_frame_id", "hdr_n_tracked", "hdr_n_lost", "hdr_n_removed_total"):
                s[k] = int(raw[Lo[k]:Lo[k] + 4].view(np.int32)[0])
            self._snap_cache = s
        return self._snap_cache

    def _views(self, list_name, n_name):
        s = self._snapshot()
        out = []
        for slot in s[list_name][:s[n_name]]:
            m = s["mean"][slot]
            r = m[:4].copy()
            if self.opts.kalman_format in ('default', 'strongsort'):
                r[2] *= r[3]
            r[:2] -= r[2:] / 2
            out.append(_PoolTrack(self, slot, s["tid"][slot], r, s["cls"][slot], s["score"][slot]))
        return out

    @property
    def tracked_stracks(self):
        return self._views("tracked", "hdr_n_tracked")

    @property
    def lost_stracks(self):
        return self._views("lost", "hdr_n_lost")

    @property
    def removed_stracks(self):
        """The reference keeps every removed track forever (an ever-growing list it only uses for id-membership tests); the device pool
        recycles their slots and keeps the count.  -> one placeholder per removed track (state Removed, no geometry)."""
        return [_RemovedTrack() for _ in range(self._snapshot()["hdr_n_removed_total"])]


class _RemovedTrack(BaseTrack):
    state = TrackState.Removed
    track_id = -1

    def __repr__(self):
        return 'OT_removed'


def joint_stracks(tlista, tlistb):
    exists, res = {}, []
    for t in tlista:
        exists[t.track_id] = 1
        res.append(t)
    for t in tlistb:
        if not exists.get(t.track_id, 0):
            exists[t.track_id] = 1
            res.append(t)
    return res


def sub_stracks(tlista, tlistb):
    stracks = {t.track_id: t for t in tlista}
    for t in tlistb:
        if stracks.get(t.track_id, 0):
            del stracks[t.track_id]
    return list(stracks.values())


def remove_duplicate_stracks(stracksa, stracksb):
    """basetrack.py:562-576: of every (a, b) pair closer than IoU distance 0.15 drop the younger track (ties: drop a)"""
    pdist = matching.iou_distance(stracksa, stracksb)
    pairs = np.where(pdist < 0.15)
    dupa, dupb = list(), list()
    for p, q in zip(*pairs):
        timep = stracksa[p].frame_id - stracksa[p].start_frame
        timeq = stracksb[q].frame_id - stracksb[q].start_frame
        if timep > timeq:
            dupb.append(q)
        else:
            dupa.append(p)
    resa = [t for i, t in enumerate(stracksa) if i not in dupa]
    resb = [t for i, t in enumerate(stracksb) if i not in dupb]
    return resa, resb
