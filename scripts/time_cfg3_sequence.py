"""cfg3's tracker chain alone, launch by launch: BoT-SORT (xywh Kalman, a camera-motion warp per frame), 500 objects, 80 frames per launch (y7t_tracker_step_frames)
over a long stretch of the bench's synthetic sequence -- how the time of a launch moves with the position in the sequence (lost / removed lists, ties -> literal re-solves).
python scripts/time_cfg3_sequence.py [launches=40]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolov7_tracker_amd import synth
from yolov7_tracker_amd.tracker.basetrack import BaseTrack
from yolov7_tracker_amd.tracker.botsort import BoTSORT

L = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, NOBJ = 80, 500
o = types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="botsort", img_size=1280, iou_thresh=0.5, max_tracks=2048, max_dets=1024, tracker_threads=int(os.environ.get("THREADS", "0")))
dets = synth.make_detections(L * B, NOBJ, 1280, seq_idx=0, bounce=True)
warps = synth.make_warps(L * B, seq_idx=0).reshape(-1, 6)
BaseTrack._count = 0
t = BoTSORT(o, frame_rate=30)
dd = [torch.from_numpy(d).cuda() for d in dets]
wd = torch.from_numpy(warps).cuda()
res = torch.zeros((L * B, t.cap_t + 1, 8), dtype=torch.float64, device="cuda")
tabs = [t.frames_table(dd[l * B:(l + 1) * B], [res[i] for i in range(l * B, (l + 1) * B)], [wd[i] for i in range(l * B, (l + 1) * B)]) for l in range(L)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(L + 1)]
torch.cuda.synchronize()
ev[0].record()
for l in range(L):
    t._launch_frames(tabs[l]); ev[l + 1].record()
torch.cuda.synchronize()
ms = [ev[l].elapsed_time(ev[l + 1]) for l in range(L)]
off = t._layout["hdr_prof"]
prof = t._state[off:off + 32 * 8].view(torch.int64).cpu().numpy()
rows = res[:, t.cap_t, 0].view(torch.int32)[::2].cpu().numpy() if False else None
cnt = [int(res[(l + 1) * B - 1, t.cap_t].view(torch.int32)[0]) for l in range(L)]
print("ms per 80-frame launch:", " ".join("%.1f" % m for m in ms))
print("tracks in the last frame of each launch:", cnt)
print("mean %.2f ms, first five %.2f, last five %.2f; status %d" % (np.mean(ms), np.mean(ms[:5]), np.mean(ms[-5:]), int(t._state[t._layout["hdr_status"]:t._layout["hdr_status"] + 4].view(torch.int32)[0])))
print("prof words 20..31:", prof[20:32].tolist())
