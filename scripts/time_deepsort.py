"""Time the DeepSORT frame step on the GPU with detections + appearance features resident in HBM; phase stamps of the step workgroup."""
import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import synth
from yolov7_tracker_amd.tracker.deepsort import DeepSORT
from yolov7_tracker_amd.tracker.basetrack import BaseTrack

def opts(**kw):
    o = types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280, iou_thresh=0.5)
    o.__dict__.update(kw); return o

names = ["predict+dets", "cascade", "set-difference", "apply+features", "IoU leftovers", "unconfirmed", "new tracks", "ageing", "finish"]
nobj, nf, dim = 80, 200, 512
gt = []
dets = synth.make_detections(nf, nobj, seq_idx=0, bounce=True, ground_truth=gt)
rng = np.random.default_rng(0)
base = rng.normal(0, 1, (4096, dim)).astype(np.float32)
feats = []
for d, g in zip(dets, gt):     # an embedding per detection: its object's own vector + noise (false positives: random)
    f = rng.normal(0, 1, (len(d), dim)).astype(np.float32)
    if len(g) and len(d):
        gc = np.stack([g[:, 1] + g[:, 3] / 2, g[:, 2] + g[:, 4] / 2], 1)
        dc = np.stack([(d[:, 0] + d[:, 2]) / 2, (d[:, 1] + d[:, 3]) / 2], 1)
        dist = np.abs(dc[:, None, :] - gc[None, :, :]).max(2)
        j = dist.argmin(1)
        hit = dist[np.arange(len(d)), j] < 4.0
        f[hit] = base[g[j[hit], 0].astype(int) % 4096] + 0.15 * f[hit]
    feats.append(f)
ddev = [torch.from_numpy(d).cuda() for d in dets]
fdev = [torch.from_numpy(f).cuda() for f in feats]
for threads in (64, 256):
    BaseTrack._count = 0
    t = DeepSORT(opts(tracker_threads=threads), reid_model=lambda c: None)
    for d, f in zip(ddev[:20], fdev[:20]): t._launch(d, f)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for d, f in zip(ddev[20:], fdev[20:]): t._launch(d, f)
    e1.record(); torch.cuda.synchronize()
    off = t._layout["hdr_prof"]
    p = t._state[off:off + 32 * 8].view(torch.int64).cpu().numpy()
    ph = np.diff(p[:10]) / 1e3
    print("threads=%d  %.1f us/frame (4 kernels), ids so far %d | last frame kcycles: " % (threads, e0.elapsed_time(e1) * 1e3 / (nf - 20), BaseTrack._count)
          + ", ".join("%s %.0f" % (n, v) for n, v in zip(names, ph)) + " | frames with several ages %d, of them contested %d, solved jointly %d" % tuple(p[27:30]))
