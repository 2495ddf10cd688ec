"""Rank sort + kept-list NMS alone (y7t_det_postprocess on predecoded candidates) at ONE frame and at 40: HIP events around the post-processing of a fused forward
whose Detect epilogues left ~2000 candidates per frame (the bench's configuration: conditioned weights, planted objectness, all four levels live).
    python scripts/time_nms.py"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from yolov7_tracker_amd import synth
from yolov7_tracker_amd.detector import arch, model

for B in (1, 40):
    frames_host = synth.make_frames(B, 80, 1280, seq_idx=0)
    sd = bench.conditioned_state_dict(types.SimpleNamespace(arch="yolov7-w6", img=1280), 10, frames_host)
    det = model.Detector(arch.yolov7_w6(10), sd, img_size=(1280, 1280), max_batch=B, seed=0)
    frames = torch.from_numpy(frames_host).cuda()
    bench.plant_objectness_bias(det, frames)
    out = det.forward(frames, fuse_decode=0.01)
    d, n = det.postprocess(out, 0.01, 0.45, None)
    torch.cuda.synchronize()
    cn = det.candidate_arrays(out.pset)[4][:B].cpu().numpy()
    keep0 = det.plan.post[out.pset].keep[:B].clone()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for e0, e1 in ev:
        e0.record(); det.postprocess(out, 0.01, 0.45, None); e1.record()
    torch.cuda.synchronize()
    us = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    assert torch.equal(keep0, det.plan.post[out.pset].keep[:B])
    print("B=%2d  candidates per frame %d..%d  kept %s  rank sort + NMS: median %.1f us, min %.1f us per launch pair" % (B, cn.min(), cn.max(), n[:B].cpu().numpy()[:4], us[len(us) // 2], us[0]), flush=True)
    del det
