"""Time the ReID branch of config 4 on the GPU: crops from a uint8 frame resident in HBM -> OSNet x0_25 -> (N, 512) features."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import synth
from yolov7_tracker_amd.tracker.reid import ReIDExtractor

frames = torch.from_numpy(synth.make_frames(1, 80, 1280, seq_idx=0)).cuda()
rng = np.random.default_rng(0)
for n in (80, 500, 2560):
    ex = ReIDExtractor(None, max_crops=n)
    xy = rng.uniform(0, 1100, (n, 2)); wh = rng.uniform(16, 160, (n, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    f = ex.features_for_boxes(frames[0], boxes); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): f = ex.features_for_boxes(frames[0], boxes)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("crops=%d  %.3f ms per call  %.1f us per crop  (%s)" % (n, ms, ms * 1e3 / n, os.environ.get("Y7T_REID_FUSED", "default")))
