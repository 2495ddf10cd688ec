"""Time the fused tracker step on the GPU: per-frame latency (launch + D2H of the returned rows) and
device-side kernel time with detections resident in HBM (no host sync inside the loop)."""
import sys, time, types
import numpy as np, torch
sys.path.insert(0, ".")
from yolov7_tracker_amd import synth, _lib
from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
from yolov7_tracker_amd.tracker.basetrack import BaseTrack

def opts(**kw):
    o = types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280, iou_thresh=0.5)
    o.__dict__.update(kw); return o

for nobj, nf in ((80, 300), (500, 60)):
    dets = synth.make_detections(nf, nobj, seq_idx=0)
    ddev = [torch.from_numpy(d).cuda() for d in dets]
    for threads in (64, 256, 1024):
        BaseTrack._count = 0
        t = ByteTrack(opts(tracker_threads=threads))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for d in ddev: t.update(d, None)
        torch.cuda.synchronize(); lat = (time.perf_counter() - t0) / nf
        BaseTrack._count = 0
        t = ByteTrack(opts(tracker_threads=threads))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for d in ddev: t._launch(d)
        e1.record(); torch.cuda.synchronize()
        print("n_obj=%d threads=%d  update() latency %.1f us/frame   kernel-only %.1f us/frame" % (nobj, threads, lat * 1e6, e0.elapsed_time(e1) * 1e3 / nf))
