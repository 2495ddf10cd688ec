"""Batch-1 latency mode with the reference's Timer semantics (tracker/track.py:140-174): wall time from "frame tensor on the
host" to "track list produced", one frame at a time, device sync at the end of every frame (tracker.update copies the rows back).
Reports fps for (a) float32 CHW host frames exactly like the reference loader hands over, (b) uint8 HWC host frames,
(c) uint8 frames already resident in HBM; each eager and with the detector+NMS chain replayed as a hipGraph."""
import sys, time, types
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import synth
from yolov7_tracker_amd.detector import arch, model
from yolov7_tracker_amd.tracker.basetrack import BaseTrack
from yolov7_tracker_amd.tracker.bytetrack import ByteTrack

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
det = model.Detector(arch.yolov7_w6(10), None, img_size=(1280, 1280), max_batch=1, seed=0)
frames = synth.make_frames(8, 80, 1280, 0)
dets_seq = synth.make_detections(N + 10, 80, 1280, 0)
opts = types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280, iou_thresh=0.5)
f32 = [(torch.from_numpy(np.ascontiguousarray(f[:, :, ::-1].transpose(2, 0, 1))).float() / 255.0).pin_memory() for f in frames]
u8 = [torch.from_numpy(f).pin_memory() for f in frames]
u8d = [t.cuda() for t in u8]
dev_in = torch.empty((1, 1280, 1280, 3), dtype=torch.uint8, device="cuda")
dev_in_f = torch.empty((1, 3, 1280, 1280), dtype=torch.float32, device="cuda")


def run(mode, graph):
    BaseTrack._count = 0
    trk = ByteTrack(opts)
    g = None
    if graph:
        src = dev_in_f if mode == "f32" else dev_in
        g, _, _ = det.capture(src, 0.01, 0.45, None)
    tot = 0.0
    for i in range(N + 10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "f32":
            x = f32[i % 8]
            if g is not None:
                dev_in_f.copy_(x[None], non_blocking=True); g.replay()
            else:
                out = det(x[None].cuda(non_blocking=True))[0]; det.postprocess(out, 0.01, 0.45, None)
        else:
            x = u8[i % 8] if mode == "u8" else u8d[i % 8]
            if g is not None:
                dev_in.copy_(x[None], non_blocking=True); g.replay()
            else:
                out = det(x[None].cuda(non_blocking=True) if mode == "u8" else x[None])[0]; det.postprocess(out, 0.01, 0.45, None)
        cur = trk.update(dets_seq[i], None)          # the scene's detections (random weights cannot detect), rows copied back
        n = sum(1 for c in cur if c.tlwh[2] * c.tlwh[3] > 150)
        dt = time.perf_counter() - t0
        if i >= 10:
            tot += dt
    return N / tot


ONLY = os.environ.get("ONLY")      # e.g. "u8,1": that combination only (scripts/latency_trace.sh)
for mode in ("f32", "u8", "u8_resident"):
    for graph in (0, 1):
        if ONLY and ONLY != "%s,%d" % (mode, graph): continue
        print("latency mode  input=%-12s hipgraph=%d  ->  %.1f fps (%.2f ms/frame)" % (mode, graph, run(mode, graph), 1e3 / run(mode, graph)))
