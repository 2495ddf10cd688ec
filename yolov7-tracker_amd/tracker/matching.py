"""Association ops with the reference's signatures (/root/reference/tracker/matching.py:30-82),
computed on the MI355X by liby7t.so: `iou_distance` (IoU with the +1 pixel convention of
cython_bbox.bbox_overlaps) and `linear_assignment` (lap.lapjv(extend_cost=True, cost_limit=t))."""
import numpy as np
import torch

from .. import _lib


def _dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=dtype)).cuda()


def ious(atlbrs, btlbrs):
    """matching.py:44-61 -> (N, M) float64 IoU."""
    return 1.0 - _cost(atlbrs, btlbrs) if (len(atlbrs) and len(btlbrs)) else np.zeros((len(atlbrs), len(btlbrs)))


def _cost(atlbrs, btlbrs):
    _lib.require_gpu()
    a = _dev(np.asarray(atlbrs, dtype=np.float64).reshape(-1, 4), np.float64)
    b = _dev(np.asarray(btlbrs, dtype=np.float64).reshape(-1, 4), np.float64)
    n, m = a.shape[0], b.shape[0]
    out = torch.empty((n, m), dtype=torch.float64, device="cuda")
    _lib.check(_lib.load().y7t_iou_cost_f64(_lib.ptr(a), n, _lib.ptr(b), m, _lib.ptr(out), _lib.stream_ptr()))
    return out.cpu().numpy()


def iou_distance(atracks, btracks):
    """matching.py:64-82: lists of tracks (objects with .tlbr) or of tlbr arrays -> (N, M) float64 cost."""
    if (len(atracks) > 0 and isinstance(atracks[0], np.ndarray)) or (len(btracks) > 0 and isinstance(btracks[0], np.ndarray)):
        atlbrs, btlbrs = atracks, btracks
    else:
        atlbrs = [t.tlbr for t in atracks]
        btlbrs = [t.tlbr for t in btracks]
    if len(atlbrs) == 0 or len(btlbrs) == 0:
        return np.zeros((len(atlbrs), len(btlbrs)), dtype=np.float64)
    return _cost(atlbrs, btlbrs)


def lapjv_device(cost, cost_limit):
    """-> (opt, x, y) like lap.lapjv(cost, extend_cost=True, cost_limit=cost_limit)."""
    _lib.require_gpu()
    L = _lib.load()
    c = _dev(cost, np.float64)
    n, m = c.shape
    x = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    y = torch.empty(max(m, 1), dtype=torch.int32, device="cuda")
    opt = torch.zeros(1, dtype=torch.float64, device="cuda")
    ws = torch.empty(max(int(L.y7t_lapjv_workspace_bytes(n, m)), 8), dtype=torch.uint8, device="cuda")
    _lib.check(L.y7t_lapjv_f64(_lib.ptr(c), n, m, float(cost_limit), _lib.ptr(x), _lib.ptr(y), _lib.ptr(opt), _lib.ptr(ws),
                               _lib.stream_ptr()))
    return float(opt.item()), x[:n].cpu().numpy().astype(np.int64), y[:m].cpu().numpy().astype(np.int64)


def lapjv_host(cost, cost_limit):
    """the same for a numpy cost matrix through the library's host-pointer entry (y7t_lapjv_f64_host: staging buffers owned by the library, one call,
    one synchronisation) -- what matching.linear_assignment uses, since the reference calls it with numpy arrays"""
    _lib.require_gpu()
    L = _lib.load()
    c = np.ascontiguousarray(cost, dtype=np.float64)
    n, m = c.shape
    x, y, opt = np.empty(max(n, 1), np.int32), np.empty(max(m, 1), np.int32), np.zeros(1, np.float64)
    _lib.check(L.y7t_lapjv_f64_host(c.ctypes.data, n, m, float(cost_limit), x.ctypes.data, y.ctypes.data, opt.ctypes.data, _lib.stream_ptr()))
    return float(opt[0]), x[:n].astype(np.int64), y[:m].astype(np.int64)


def linear_assignment(cost_matrix, thresh):
    """matching.py:30-41 -> (matches (K,2) int, unmatched_a, unmatched_b)."""
    on_device = torch.is_tensor(cost_matrix) and cost_matrix.device.type == "cuda"      # (the reference only ever passes numpy; a device cost stays there)
    if not on_device:
        cost_matrix = np.asarray(cost_matrix.cpu() if torch.is_tensor(cost_matrix) else cost_matrix)
    if (cost_matrix.numel() if on_device else cost_matrix.size) == 0:
        return np.empty((0, 2), dtype=int), tuple(range(cost_matrix.shape[0])), tuple(range(cost_matrix.shape[1]))
    if on_device:
        _, x, y = lapjv_device(cost_matrix, thresh)
    else:
        _, x, y = lapjv_host(cost_matrix, thresh)
    matches = np.asarray([[ix, mx] for ix, mx in enumerate(x) if mx >= 0])
    return matches, np.where(x < 0)[0], np.where(y < 0)[0]
