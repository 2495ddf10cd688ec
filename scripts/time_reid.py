"""Time the ReID branch of config 4 on the GPU: crops from a uint8 frame resident in HBM -> OSNet x0_25 -> (N, 512) features."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import synth
from yolov7_tracker_amd.tracker.reid import ReIDExtractor

frames = torch.from_numpy(synth.make_frames(1, 80, 1280, seq_idx=0)).cuda()
rng = np.random.default_rng(0)
for n in (80, 500, 2560):
    ex = ReIDExtractor(None, max_crops=n, fused=os.environ.get("Y7T_REID_FUSED", "1") != "0")
    xy = rng.uniform(0, 1100, (n, 2)); wh = rng.uniform(16, 160, (n, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    bdev = torch.from_numpy(boxes).cuda()
    idx = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(20): f = ex.features_for_frames(frames, bdev, idx)      # warm-up long enough for the clocks to come up
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(20, int(300 / (0.2 + n / 1500.0)))                           # ~0.3 s of kernel time
    e0.record()
    for _ in range(reps): f = ex.features_for_frames(frames, bdev, idx)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("crops=%d  %.3f ms per call  %.2f us per crop  (%d reps; %s)" % (n, ms, ms * 1e3 / n, reps, "fused MFMA kernel" if ex.fused else "fp32 op list"))
