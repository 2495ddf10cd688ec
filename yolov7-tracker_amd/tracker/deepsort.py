"""DeepSORT (/root/reference/tracker/deepsort.py:10-227) on the device track pool: the matching cascade over the gated appearance
cost (nearest cosine distance to a track's last 100 appearance vectors, Mahalanobis gate on the predicted Kalman state), the IoU
fallbacks and the list bookkeeping run in liby7t.so (y7t_tracker_step_deepsort; csrc/y7t_track_deepsort.h).

Appearance features enter at the reference's own seam, `get_feature(tlbrs, ori_img) -> (N, D)` (deepsort.py:19-41): by default it
crops `ori_img` like the reference and calls `self.reid_model(crops)`; `reid_model` is any callable returning (N, D) features -- the
device ReID extractor of this package (`tracker/reid.py`), or a stand-in.  The reference hard-wires `Extractor(opts.reid_model_path)`
with weights/ckpt.t7, which does not ship with it."""
import os

import numpy as np
import torch

from .. import _lib
from .basetrack import BaseTracker

STORE_FEATURES_BUDGET = 100     # STrack.__init__ store_features_budget (basetrack.py:76)


class DeepSORT(BaseTracker):
    _KIND = 3  # Y7T_TRACKER_DEEPSORT

    def __init__(self, opts, frame_rate=30, gamma=0.02, reid_model=None, *args, **kwargs):
        if getattr(opts, "kalman_format", "default") not in ("default", "strongsort"):
            raise NotImplementedError("DeepSORT gates on xyah measurements (deepsort.py:59): kalman_format default / strongsort")
        super().__init__(opts, frame_rate=frame_rate)
        self.reid_model = reid_model if reid_model is not None else getattr(opts, "reid_model", None)
        path = getattr(opts, "reid_model_path", None)
        if self.reid_model is None and isinstance(path, str) and path.startswith("random"):
            # "random[:osnet|:deepsort]" -- seeded random weights of the named embedding network, like the detector's "random:<arch>" model
            # paths: for synthetic runs (track.py --dataset synthetic, bench.py); no checkpoint ships that the reference's DeepSORT can load
            from .reid import ReIDExtractor
            arch = path.partition(":")[2] or "osnet"
            if arch not in ("osnet", "deepsort"):
                raise ValueError("reid_model_path %r: random, random:osnet or random:deepsort" % (path,))
            self.reid_model = ReIDExtractor(None, arch=arch, max_crops=512 if arch == "osnet" else 128)
        elif self.reid_model is None and path and os.path.isfile(str(path)):      # deepsort.py:14: Extractor(opts.reid_model_path)
            from .reid import ReIDExtractor
            self.reid_model = ReIDExtractor.from_checkpoint(path)
        self.gamma = gamma
        self.filter_small_area = False
        self._feat = None           # feature state, allocated when the feature dimension is known
        self._feat_dim = 0
        self._feat_used = False     # a frame with appearance vectors has been stepped (from then on the width is fixed)

    def get_feature(self, tlbrs, ori_img):
        """deepsort.py:19-41: crops of the boxes -> self.reid_model(crops) -> (N, D) features"""
        if self.reid_model is None:
            raise _lib.Y7TError("DeepSORT needs appearance features: pass reid_model=<callable(list of crops) -> (N, D)> (e.g. "
                                "yolov7_tracker_amd.tracker.reid.ReIDExtractor) or override get_feature")
        if hasattr(self.reid_model, "features_for_boxes"):      # device extractor: crop + resize + normalise on the GPU
            return self.reid_model.features_for_boxes(ori_img, tlbrs)
        if isinstance(ori_img, torch.Tensor):
            ori_img = ori_img.cpu().numpy()
        crops = []
        for tlbr in tlbrs:
            x1, y1, x2, y2 = (int(v) for v in tlbr)
            crops.append(ori_img[y1:y2, x1:x2])
        return self.reid_model(crops) if crops else np.zeros((0, max(self._feat_dim, 1)), np.float32)

    def _ensure_feature_state(self, dim):
        """the per-slot appearance rings + per-frame scratch, sized for `dim`-wide embeddings.  Until a frame has carried a detection above det_thresh
        no track exists (a new track needs score > det_thresh + 0.1, deepsort.py:207) and no slot holds a vector, so a state that was sized on a
        guess for such frames (empty / low-confidence first frames are common in real footage) is simply re-made when the real width shows up."""
        if self._feat is None or (int(dim) != self._feat_dim and not self._feat_used):
            self._feat_dim = int(dim)
            nb = int(self._L.y7t_deepsort_feature_bytes(self.cap_t, self.cap_d, self._feat_dim, STORE_FEATURES_BUDGET))
            self._feat = torch.zeros(nb, dtype=torch.uint8, device="cuda")
            _lib.check(self._L.y7t_deepsort_init(_lib.ptr(self._feat), nb, self.cap_t, self.cap_d, self._feat_dim, STORE_FEATURES_BUDGET,
                                                 _lib.stream_ptr()))
        elif int(dim) != self._feat_dim:
            raise ValueError("feature dimension changed from %d to %d" % (self._feat_dim, int(dim)))

    def _check_feats(self, feats_dev, n):
        """what the device step dereferences: n rows of `_feat_dim` contiguous float32 on the GPU (a short, fp16 or strided tensor would be read out
        of bounds / misinterpreted by k_ds_normalize, k_embed_dist and k_ds_store)"""
        if not isinstance(feats_dev, torch.Tensor) or feats_dev.dim() != 2:
            raise _lib.Y7TError("DeepSORT: features must be an (n, D) tensor")
        if feats_dev.shape[0] < n:
            raise _lib.Y7TError("DeepSORT: %d feature rows for %d detections" % (feats_dev.shape[0], n))
        return feats_dev.to(device="cuda", dtype=torch.float32).contiguous()

    def _launch(self, det_dev, feats_dev=None, out=None, **kw):
        """enqueue one frame step without a host round trip (pipelines / bench.py): det_dev (n, 6) float32 and feats_dev (n, D) float32
        DEVICE tensors (rows at or below det_thresh are ignored by the step), out like BaseTracker._launch.  det_dev None: the predict-only
        step of update_without_detection (basetrack.py:489-537), the same for every tracker."""
        if det_dev is None:
            return super()._launch(None, out=out, **kw)
        if feats_dev is None:
            raise _lib.Y7TError("DeepSORT._launch needs the detections' appearance features (use update() for the get_feature seam)")
        d = det_dev.reshape(-1, 6)
        n = d.shape[0]
        if n > self.cap_d:
            raise _lib.Y7TError("%d detections exceed the pool capacity max_dets=%d" % (n, self.cap_d))
        d = d.to(device="cuda", dtype=torch.float32).contiguous()
        feats_dev = self._check_feats(feats_dev, n)
        self._ensure_feature_state(feats_dev.shape[1])
        self._feat_used = self._feat_used or n > 0
        self._det_keep = (d, feats_dev)
        if out is None:
            optr, cptr = _lib.ptr(self._out), self._count_ptr
        else:
            import ctypes
            optr, cptr = _lib.ptr(out), ctypes.c_void_p(out.data_ptr() + self.cap_t * 8 * 8)
        _lib.check(self._L.y7t_tracker_step_deepsort(_lib.ptr(self._state), _lib.ptr(self._feat), self.cap_t, _lib.ptr(d), n, _lib.ptr(feats_dev),
                                                     optr, self.cap_t, cptr, self.threads, _lib.stream_ptr()))
        self.frame_id += 1
        self._snap_cache = None

    def update(self, det_results, ori_img=None):
        """(N,6) [x1,y1,x2,y2,conf,cls] + the frame -> list of tracks (deepsort.py:79-227)"""
        if isinstance(det_results, torch.Tensor):
            det_host = det_results.detach().cpu().numpy()
        else:
            det_host = np.asarray(det_results)
        det_host = np.ascontiguousarray(det_host, dtype=np.float32).reshape(-1, 6)
        n = det_host.shape[0]
        if n > self.cap_d:
            raise _lib.Y7TError("%d detections exceed the pool capacity max_dets=%d" % (n, self.cap_d))
        keep = det_host[:, 4] > np.float32(self.det_thresh)            # deepsort.py:98: only these get features
        feats = None
        if keep.any():
            feats = self.get_feature(det_host[keep, :4], ori_img)
            if not isinstance(feats, torch.Tensor):
                feats = torch.from_numpy(np.ascontiguousarray(feats, dtype=np.float32))
            feats = self._check_feats(feats, int(keep.sum()))
            self._ensure_feature_state(feats.shape[1])
            self._feat_used = True
        elif self._feat is None:        # nothing above det_thresh yet: the extractor's width if it states one, else a placeholder that the first real frame replaces
            self._ensure_feature_state(getattr(self.reid_model, "feat_dim", None) or self._feat_dim or 128)
        d = torch.from_numpy(det_host).cuda()
        allf = torch.zeros((max(n, 1), self._feat_dim), dtype=torch.float32, device="cuda")
        if feats is not None:
            allf[torch.from_numpy(np.nonzero(keep)[0]).cuda()] = feats
        self._det_keep = (d, allf)
        _lib.check(self._L.y7t_tracker_step_deepsort(_lib.ptr(self._state), _lib.ptr(self._feat), self.cap_t, _lib.ptr(d), n, _lib.ptr(allf),
                                                     _lib.ptr(self._out), self.cap_t, self._count_ptr, self.threads, _lib.stream_ptr()))
        self.frame_id += 1
        self._snap_cache = None
        rows = self._collect()
        st = int(self._feat[20:24].view(torch.int32).item())            # Y7TFeatHdr.status
        if st:
            raise _lib.Y7TError("DeepSORT feature state overflow (status %d)" % st)
        return rows
