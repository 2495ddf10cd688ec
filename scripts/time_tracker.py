"""Time the fused tracker step on the GPU: per-frame latency (launch + D2H of the returned rows) and
device-side kernel time with detections resident in HBM (no host sync inside the loop)."""
import sys, time, types
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import synth, _lib
from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
from yolov7_tracker_amd.tracker.basetrack import BaseTrack

def opts(**kw):
    o = types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280, iou_thresh=0.5)
    o.__dict__.update(kw); return o

for nobj, nf in ((80, 300), (500, 60)):
    dets = synth.make_detections(nf, nobj, seq_idx=0)
    ddev = [torch.from_numpy(d).cuda() for d in dets]
    for threads in (64, 256, 512, 1024):
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

# phase breakdown of one step (shader-clock stamps written by thread 0)
names = ["setup+lists", "multi_predict", "dets+gather", "assoc1 (IoU+LAP)", "apply1", "lists2", "assoc2", "apply2+lists", "assoc3", "apply3+new+age", "finish"]
for nobj in (80, 500):
    dets = synth.make_detections(40, nobj, seq_idx=0)
    BaseTrack._count = 0
    t = ByteTrack(opts(max_tracks=1024, max_dets=1024))
    acc = np.zeros(11)
    for i, d in enumerate(dets):
        t.update(d, None)
        off = t._layout["hdr_prof"]
        p = t._state[off:off + 32 * 8].view(torch.int64).cpu().numpy()
        if i >= 10:
            acc += np.diff(p[:12])
    acc /= (len(dets) - 10)
    print("n_obj=%d  total %.0f kcycles: " % (nobj, acc.sum() / 1e3) + ", ".join("%s %.0f" % (n, v / 1e3) for n, v in zip(names, acc)))
    sp = np.diff(p[16:22]) / 1e3
    print("   sparse association #1 (kcycles, last frame): cost pass %.0f, candidate sort %.0f, forced decisions %.0f, components %.0f, per-component solves %.0f" % tuple(sp))
    if p[23]:      # wave 0's share of the per-component solves (y7t_assoc_sparse_try step 4a; prof[23] = components | rows << 16 | columns << 24 | ticket << 32 of wave 0's last one)
        print("      components of two or more rows: %d, each on a wave; wave 0 done after %.0f kcycles; its last component (ticket %d): %d rows x %d columns, %d search steps"
              % (p[23] & 0xffff, (p[22] - p[20]) / 1e3, (p[23] >> 32) & 0xffff, (p[23] >> 16) & 255, (p[23] >> 24) & 255, p[29]))
