"""Time the DeepSORT frame step on the GPU (library named by Y7T_LIB): the reference-recorded crowd sequence through DeepSORT._launch (detections and features resident
on the device, no host sync between frames), kernel-only time per frame and the shader-clock stamps of the step's phases (csrc/y7t_track_deepsort.h: Y7T_PROF / Y7T_CPROF)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import util
from yolov7_tracker_amd.tracker.basetrack import BaseTrack
from yolov7_tracker_amd.tracker.deepsort import DeepSORT

for name in ("deepsort_crowd", "deepsort_dim512"):
    trk, fmt, dets, want = util.load_tracker_case(name)
    feat = util.feature_fn_for(name)
    ddev = [torch.from_numpy(d).cuda() for d in dets]
    fdev = [torch.from_numpy(feat(d[:, :4])).cuda() for d in dets]
    for rep in range(2):
        BaseTrack._count = 0
        t = DeepSORT(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format=fmt, img_size=1280, iou_thresh=0.5))
        res = torch.zeros((len(dets), t.cap_t + 1, 8), dtype=torch.float64, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        acc, accc = np.zeros(9), np.zeros(4)
        for i in range(len(dets)):
            t._launch(ddev[i], fdev[i], out=res[i])
            if rep == 1:
                off = t._layout["hdr_prof"]
                p = t._state[off:off + 32 * 8].view(torch.int64).cpu().numpy()
                if i >= 5:
                    acc += np.diff(p[:10]); accc += p[16:20]
        e1.record(); torch.cuda.synchronize()
        if rep == 0:
            print("%s %s: %d frames, %.1f detections per frame, kernel-only %.1f us/frame" % (os.path.basename(os.environ.get("Y7T_LIB", "liby7t.so")), name, len(dets),
                  np.mean([len(d) for d in dets]), e0.elapsed_time(e1) * 1e3 / len(dets)), flush=True)
        else:
            n = len(dets) - 5
            print("   phases (kcycles per frame, stamps 0..9): " + ", ".join("%.0f" % (v / n / 1e3) for v in acc) + " | cascade parts (prof 16..19): " + ", ".join("%.0f" % (v / n / 1e3) for v in accc), flush=True)
