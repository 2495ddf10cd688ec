"""BoT-SORT state path (/root/reference/tracker/botsort.py:250-493) on the device pool: xywh Kalman filter
(BoTSORTKalmanFilter, kalman_filter.py:414-605), camera-motion compensation of the predicted tracks (multi_gmc,
botsort.py:250-269) and BoT-SORT's association variant -- including its two quirks (every unmatched pool track, not only the
Tracked ones, goes to the low-score association, botsort.py:411; new tracks are spawned from the detections left after the
FIRST association, botsort.py:462-466).

Out of scope: the camera-motion ESTIMATION (`GMC`, OpenCV ORB / RANSAC, botsort.py:13-248) and the ReID
appearance branch (off by default in the reference too: use_apperance_model = False).  The 2x3 warp of a frame is an input:
`update(dets, img, warp=H)` or a user-supplied `tracker.gmc = callable(raw_frame, detections) -> H`."""
import numpy as np
import torch

from .basetrack import BaseTracker, STrack, TrackState, joint_stracks, sub_stracks  # noqa: F401


class BoTSORT(BaseTracker):
    _KIND = 2  # Y7T_TRACKER_BOTSORT

    def __init__(self, opts, frame_rate=30, gamma=0.02, use_GMC=True, *args, **kwargs):
        if getattr(opts, "kalman_format", "botsort") != "botsort":
            opts.kalman_format = "botsort"          # tracker/track.py:68-69 forces it for this tracker
        super().__init__(opts, frame_rate=frame_rate)
        self.use_apperance_model = False
        self.gamma = gamma
        self.low_conf_thresh = max(0.15, self.opts.conf_thresh - 0.3)
        self.filter_small_area = False
        self.use_GMC = use_GMC
        self.gmc = None                              # optional callable(raw_frame, detections) -> (2, 3) matrix
        self.theta_iou, self.theta_emb = 0.5, 0.25
        self._warp = torch.zeros(6, dtype=torch.float64, device="cuda")

    _warned = False

    def update(self, det_results, ori_img=None, warp=None):
        if warp is None and self.use_GMC and self.gmc is not None:
            warp = self.gmc(ori_img, det_results)
        if warp is None and self.use_GMC and not BoTSORT._warned:
            import warnings
            warnings.warn("BoTSORT: use_GMC is set but no camera-motion matrix was supplied (update(..., warp=H) or tracker.gmc = callable): "
                          "the reference estimates one per frame with OpenCV (botsort.py:13-248, out of scope here); running WITHOUT "
                          "compensation, results on moving-camera footage will differ from the reference", RuntimeWarning)
            BoTSORT._warned = True
        w = None
        if warp is not None and self.use_GMC:
            self._warp.copy_(torch.as_tensor(np.ascontiguousarray(warp, dtype=np.float64).reshape(6)), non_blocking=True)
            w = self._warp
        self._launch(det_results, warp=w)
        return self._collect()
