"""batch-1 latency probe: one frame at a time (reference Timer semantics), prints ms per frame; under rocprofv3 gives the per-kernel picture"""
import sys, time, types
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import synth
from yolov7_tracker_amd.detector import arch, model
from yolov7_tracker_amd.tracker.basetrack import BaseTrack
from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
det = model.Detector(arch.yolov7_w6(10), None, img_size=(1280, 1280), max_batch=1, seed=0)
frames = synth.make_frames(4, 80, 1280, 0)
u8 = [torch.from_numpy(f).pin_memory() for f in frames]
det.plant_objectness_bias(u8[0][None].cuda())
dets_seq = synth.make_detections(N + 10, 80, 1280, 0, bounce=True)
BaseTrack._count = 0
trk = ByteTrack(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280, iou_thresh=0.5, max_tracks=512, max_dets=512))
tot = 0.0
parts = np.zeros(4)
for i in range(N + 10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = u8[i % 4][None].cuda(non_blocking=True)
    out = det.forward(x, fuse_decode=0.01)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    det.postprocess(out, 0.01, 0.45, None)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    cur = trk.update(dets_seq[i], None)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    if i >= 10:
        tot += t3 - t0; parts += [t1 - t0, t2 - t1, t3 - t2, 0]
print("latency (with phase syncs): %.3f ms/frame  forward+h2d %.3f  nms %.3f  tracker %.3f" % (tot / N * 1e3, parts[0] / N * 1e3, parts[1] / N * 1e3, parts[2] / N * 1e3))
print("launch list:", det.launch_list(1))
