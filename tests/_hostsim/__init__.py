"""tests/_hostsim -- TEST INFRASTRUCTURE ONLY: CPU build (nt = 1) of the portable tracker
workgroup programs, so their control flow can be tested without a GPU.  Never imported by the
product package."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Y7T_HOSTSIM_DEFS="-DY7T_NEXT_TRACKER=1": a second library with experimental macros of the headers switched on (the tests then run against it)
_DEFS = os.environ.get("Y7T_HOSTSIM_DEFS", "").split()
_SO = os.path.join(_HERE, "liby7t_hostsim%s.so" % ("_" + "".join(c for c in "".join(_DEFS) if c.isalnum()) if _DEFS else ""))
_SRC = os.path.join(_HERE, "y7t_hostsim.cpp")
_CSRC = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "yolov7-tracker_amd", "csrc")


def build(force=False):
    deps = [_SRC, os.path.join(_CSRC, "y7t_track_core.h"), os.path.join(_CSRC, "y7t_track_step.h"), os.path.join(_CSRC, "y7t_track_deepsort.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"] + _DEFS + ["-o", _SO, _SRC])
    # the workgroup's fast scratch: 131072 bytes like the device's LDS budget (csrc/y7t_tracker.hip: kFastBytes), so the placement branches taken here are
    # the ones the GPU takes; Y7T_HOSTSIM_FAST_BYTES=0 runs everything out of the state blob (the other branches), any other value sizes it
    ctypes.CDLL(_SO).hs_set_fast_bytes(int(os.environ.get("Y7T_HOSTSIM_FAST_BYTES", "131072")))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.hs_tracker_bytes.restype = ctypes.c_size_t
        L.hs_tracker_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.hs_tracker_init.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_double] * 3 + [ctypes.c_void_p]
        L.hs_tracker_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.hs_kf_gmc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.hs_tracker_status.argtypes = [ctypes.c_void_p]
        L.hs_arena_begin.argtypes = [ctypes.c_void_p]
        L.hs_arena_end.argtypes = [ctypes.c_void_p]
        L.hs_lapjv.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        L.hs_lapsap.argtypes = L.hs_lapjv.argtypes
        L.hs_laplit.argtypes = L.hs_lapjv.argtypes
        L.hs_iou_cost.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.hs_kf_initiate.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.hs_kf_predict.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.hs_kf_update.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
        L.hs_kf_project.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        L.hs_kf_gating.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.hs_kf_gating.restype = ctypes.c_double
        L.hs_feat_bytes.restype = ctypes.c_size_t
        L.hs_feat_bytes.argtypes = [ctypes.c_int] * 4
        L.hs_feat_init.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4
        L.hs_deepsort_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.hs_feat_status.argtypes = [ctypes.c_void_p]
        L.hs_pyset_difference.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib = L
    return _lib


def pyset_difference(n, matched):
    """order of `list(set(range(n)) - set(matched))` as the device program emulates it"""
    member = np.zeros(max(n, 1), np.int32)
    member[list(matched)] = 1
    out = np.zeros(max(n, 1), np.int32)
    c = lib().hs_pyset_difference(n, member.ctypes.data, len(set(matched)), out.ctypes.data)
    assert c >= 0
    return out[:c].tolist()


def lapjv(cost, limit, sap=False, literal=False):
    """JV on lap's implicit extended matrix (sap=False), the reduced shortest-augmenting-path solver with its tie fallback (sap=True), or
    lapjv.cpp run literally (literal=True)"""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    x = np.empty(max(nr, 1), np.int32)
    y = np.empty(max(nc, 1), np.int32)
    if nr and nc:
        (lib().hs_laplit if literal else lib().hs_lapsap if sap else lib().hs_lapjv)(cost.ctypes.data, nr, nc, float(limit), x.ctypes.data, y.ctypes.data)
    else:
        x[:] = -1
        y[:] = -1
    return x[:nr].copy(), y[:nc].copy()


class HostSimTracker:
    TRACKERS = {"sort": 0, "bytetrack": 1, "botsort": 2, "deepsort": 3}
    KINDS = {"default": 0, "naive": 1, "botsort": 2, "strongsort": 3}

    def __init__(self, kind="bytetrack", conf_thresh=0.2, track_buffer=30, kalman_format="default", iou_thresh=0.5,
                 frame_rate=30, cap_t=1024, cap_d=1024, ids=None, f32_quirk=1, feature_fn=None, feat_dim=128, feat_budget=100):
        self.ids = ids if ids is not None else np.zeros(1, np.int32)
        n = lib().hs_tracker_bytes(cap_t, cap_d)
        self.blob = np.zeros(n, np.uint8)
        self.cap_t = cap_t
        lib().hs_tracker_init(self.blob.ctypes.data, self.TRACKERS[kind], self.KINDS[kalman_format], cap_t, cap_d,
                              int(frame_rate / 30.0 * track_buffer), f32_quirk, conf_thresh, max(0.15, conf_thresh - 0.3),
                              iou_thresh, self.ids.ctypes.data)
        self.out = np.zeros((cap_t, 8), np.float64)
        self.kind, self.feature_fn, self.feat_dim = kind, feature_fn, feat_dim
        if kind == "deepsort":
            self.fblob = np.zeros(lib().hs_feat_bytes(cap_t, cap_d, feat_dim, feat_budget), np.uint8)
            lib().hs_feat_init(self.fblob.ctypes.data, cap_t, cap_d, feat_dim, feat_budget)

    def update(self, det, warp=None):
        if self.kind == "deepsort" and det is not None:
            det = np.ascontiguousarray(det, dtype=np.float32).reshape(-1, 6)
            feats = np.zeros((max(len(det), 1), self.feat_dim), np.float32)
            if len(det):
                feats[:len(det)] = self.feature_fn(det[:, :4])     # (the reference extracts them for the rows above det_thresh only)
            cnt = lib().hs_deepsort_step(self.blob.ctypes.data, self.fblob.ctypes.data, det.ctypes.data, det.shape[0], feats.ctypes.data,
                                         self.out.ctypes.data, self.cap_t)
            if lib().hs_tracker_status(self.blob.ctypes.data) or lib().hs_feat_status(self.fblob.ctypes.data):
                raise RuntimeError("tracker capacity exceeded")
            return [(int(r[0]), r[1:5].copy(), float(r[5]), float(r[6])) for r in self.out[:cnt]]
        wp = None
        if warp is not None:
            self._warp = np.ascontiguousarray(warp, dtype=np.float64).reshape(6)
            wp = self._warp.ctypes.data
        if det is None:
            n, ptr = -1, None
        else:
            det = np.ascontiguousarray(det, dtype=np.float32).reshape(-1, 6)
            n, ptr = det.shape[0], det.ctypes.data
        cnt = lib().hs_tracker_step(self.blob.ctypes.data, ptr, n, self.out.ctypes.data, self.cap_t, wp)
        st = lib().hs_tracker_status(self.blob.ctypes.data)
        if st:
            raise RuntimeError("tracker capacity exceeded (status %d)" % st)
        return [(int(r[0]), r[1:5].copy(), float(r[5]), float(r[6])) for r in self.out[:cnt]]


def run(kind, dets_per_frame, warps=None, arena_frames=0, **kw):
    """arena_frames > 0: the frames run in groups of that many with the index lists in the launch-long arena (y7t_arena_load ... frames ... y7t_arena_store),
    like the frames of one k_tracker_step_frames launch on the device"""
    trk = HostSimTracker(kind, **kw)
    out = []
    for i, d in enumerate(dets_per_frame):
        if arena_frames and i % arena_frames == 0:
            assert lib().hs_arena_begin(trk.blob.ctypes.data)
        out.append(trk.update(d, None if warps is None else warps[i]))
        if arena_frames and (i % arena_frames == arena_frames - 1 or i == len(dets_per_frame) - 1):
            lib().hs_arena_end(trk.blob.ctypes.data)
    return out
