"""oracle/cnative.py -- TEST INFRASTRUCTURE ONLY.

ctypes bindings for oracle/liby7t_oracle.so (the plain-C restatement of the
third-party kernels `lap.lapjv`, `cython_bbox.bbox_overlaps` and
`torchvision.ops.nms`; see y7t_oracle.c for citations and the "parity unpinned"
note).  Exposes them with the SAME Python signatures the reference calls
(/root/reference/tracker/matching.py:34,56-59; utils/general.py:679) so they can be
injected as stub modules `lap` / `cython_bbox` when the reference's own tracker
sources are imported by oracle/ref_harness.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liby7t_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "y7t_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        fp = ctypes.POINTER(ctypes.c_float)
        L.y7o_lapjv_square.argtypes = [ctypes.c_int, dp, ip, ip]
        L.y7o_lapjv_square.restype = None
        L.y7o_lapjv_extend.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ip, ip]
        L.y7o_lapjv_extend.restype = ctypes.c_double
        L.y7o_bbox_overlaps.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp]
        L.y7o_bbox_overlaps.restype = None
        L.y7o_nms_f32.argtypes = [fp, ip, ctypes.c_int, ctypes.c_float, ip]
        L.y7o_nms_f32.restype = ctypes.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def lapjv(cost, extend_cost=False, cost_limit=np.inf, return_cost=True):
    """Restatement of lap.lapjv's Python wrapper semantics (lap/_lapjv.pyx)."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    if cost.ndim != 2:
        raise ValueError("2-dimensional array expected")
    nr, nc = cost.shape
    if cost_limit < np.inf:
        x = np.empty(max(nr, 1), dtype=np.int32)
        y = np.empty(max(nc, 1), dtype=np.int32)
        opt = lib().y7o_lapjv_extend(_dp(cost), nr, nc, float(cost_limit), _ip(x), _ip(y))
        x, y = x[:nr], y[:nc]
    else:
        if nr != nc:
            if not extend_cost:
                raise ValueError("Square cost array expected; pass extend_cost=True")
            n = max(nr, nc)
            ext = np.zeros((n, n), dtype=np.float64)
            ext[:nr, :nc] = cost
        else:
            n, ext = nr, cost
        xe = np.empty(max(n, 1), dtype=np.int32)
        ye = np.empty(max(n, 1), dtype=np.int32)
        lib().y7o_lapjv_square(n, _dp(np.ascontiguousarray(ext)), _ip(xe), _ip(ye))
        x, y = xe[:nr].copy(), ye[:nc].copy()
        x[x >= nc] = -1
        y[y >= nr] = -1
        opt = float(cost[np.nonzero(x >= 0)[0], x[x >= 0]].sum())
    x = x.astype(np.int64)
    y = y.astype(np.int64)
    return (opt, x, y) if return_cost else (x, y)


def bbox_overlaps(boxes, query_boxes):
    """cython_bbox.bbox_overlaps(boxes (N,4) f64, query (K,4) f64) -> (N,K) f64."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 4)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float64).reshape(-1, 4)
    n, k = boxes.shape[0], query_boxes.shape[0]
    out = np.zeros((n, k), dtype=np.float64)
    if n and k:
        lib().y7o_bbox_overlaps(_dp(boxes), n, _dp(query_boxes), k, _dp(out))
    return out


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms greedy semantics on numpy float32 arrays -> kept indices
    (score-descending; ties broken by original index ascending)."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.lexsort((np.arange(n), -scores.astype(np.float64))).astype(np.int32)
    keep = np.empty(n, dtype=np.int32)
    nk = lib().y7o_nms_f32(boxes.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _ip(order), n,
                           ctypes.c_float(iou_threshold), _ip(keep))
    return keep[:nk].astype(np.int64)
