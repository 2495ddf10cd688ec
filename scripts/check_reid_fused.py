"""fused OSNet kernel vs the fp32 op-list executor (same crops from the same frame), error statistics + timing"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import synth
from yolov7_tracker_amd.tracker.reid import ReIDExtractor

frames = torch.from_numpy(synth.make_frames(2, 80, 1280, seq_idx=0)).cuda()
rng = np.random.default_rng(0)
n = 200
xy = rng.uniform(0, 1100, (n, 2)); wh = rng.uniform(16, 160, (n, 2))
boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
boxes[0] = [-20, -10, 40, 90]; boxes[1] = [1250, 1200, 1400, 1300]; boxes[2] = [5, 5, 5, 50]      # clipped / empty
a = ReIDExtractor(None, seed=3, max_crops=4096, fused=False)
b = ReIDExtractor(None, seed=3, max_crops=4096, fused=True)
fa = a.features_for_boxes(frames[0], boxes).cpu().numpy()
fb = b.features_for_boxes(frames[0], boxes).cpu().numpy()
err = np.abs(fa - fb)
cos = (fa * fb).sum(1) / (np.linalg.norm(fa, axis=1) * np.linalg.norm(fb, axis=1) + 1e-12)
print("ref |mean| %.4f max %.4f | err mean %.5f max %.5f | rel-to-max %.5f | cos min %.6f (row %d)" % (np.abs(fa).mean(), np.abs(fa).max(), err.mean(), err.max(), err.max() / np.abs(fa).max(), cos[:2].min() if False else cos.min(), int(cos.argmin())))
print("per-row max err first 5:", err.max(1)[:5], "nonfinite:", int((~np.isfinite(fb)).sum()))
idx = rng.integers(0, 2, n).astype(np.int32)
f2 = b.features_for_frames(frames, boxes, idx).cpu().numpy()
f2a = np.stack([a.features_for_boxes(frames[int(i)], boxes[k:k + 1]).cpu().numpy()[0] for k, i in enumerate(idx[:20])])
print("batch api err (first 20 rows)", np.abs(f2[:20] - f2a).max())
for ex, name in ((a, "op list fp32"), (b, "fused")):
    for m in (80, 2560):
        bx = np.tile(boxes, (m // n + 1, 1))[:m]
        ex.features_for_boxes(frames[0], bx); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ex.features_for_boxes(frames[0], bx)
        e1.record(); torch.cuda.synchronize()
        print("%-14s crops=%-5d %.3f ms per call  %.2f us per crop" % (name, m, e0.elapsed_time(e1) / 5, e0.elapsed_time(e1) / 5 * 1e3 / m))
