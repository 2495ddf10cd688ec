"""ByteTrack (/root/reference/tracker/bytetrack.py:8-204): high/low-score two-stage IoU association.
Same plugin surface as the reference; the frame step runs as one device kernel (see basetrack.py)."""
from .basetrack import BaseTracker, STrack, TrackState, joint_stracks, sub_stracks  # noqa: F401


class ByteTrack(BaseTracker):
    _KIND = 1  # Y7T_TRACKER_BYTETRACK

    def __init__(self, opts, frame_rate=30, *args, **kwargs):
        super().__init__(opts, frame_rate=frame_rate)
        # bytetrack.py:12-17: the reference also constructs an (unused) ReID Extractor here
        self.use_apperance_model = False
        self.low_conf_thresh = max(0.15, self.opts.conf_thresh - 0.3)
        self.filter_small_area = False
