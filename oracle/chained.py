"""TEST INFRASTRUCTURE (never imported by the product): the reference's per-frame loop as ONE chain on the CPU --
`model(img)[0]` -> `post_process_v7` (non_max_suppression(conf_thres=0.01) -> scale_coords -> round) -> `tracker.update(out, img0)` ->
rows kept by `min_area` -- /root/reference/tracker/track.py:138-174,234-244, tracker/bytetrack.py:41-204.

VERDICT r4 weak 1: either side of the (n, 6) hand-over was parity-tested, the seam itself was not.  This module runs the oracle's halves
back to back and grades two track streams against each other with the TrackEval-style harness."""
import os

import numpy as np
import torch

from . import detector_torch as dt
from . import tracker_np


def images(frames_host):
    """uint8 BGR HWC frames -> the loader's float32 RGB CHW tensors in [0, 1] (tracker/tracker_dataloader.py:83-88)"""
    return (torch.from_numpy(np.ascontiguousarray(frames_host[..., ::-1])).permute(0, 3, 1, 2).float() / 255.0).contiguous()


def oracle_detections(nodes, sd, anchors, frames_host, chunk=4, conf_thres=0.01, iou_thres=0.45, keep_candidates=False, fp16=False):
    """fp32 network -> decode -> NMS -> scale_coords -> round for every frame: list of float32 (n, 6) [x1, y1, x2, y2, conf, cls] (track.py:234-244);
    keep_candidates: also per frame (candidate dict by anchor row, kept anchor rows in output order) for explain_kept_set_difference.
    fp16=True: the oracle's own emulation of the device arithmetic (fp16 weights / activations, fp32 accumulate: detector_torch.forward(fp16=True)) -- the noise floor
    a chained comparison can be held to"""
    H, W = frames_host.shape[1:3]
    out, cands = [], []
    for lo in range(0, len(frames_host), chunk):
        dec, _ = dt.forward(nodes, sd, images(frames_host[lo:lo + chunk]), anchors, fp16=fp16)
        res = dt.non_max_suppression(dec, conf_thres, iou_thres)
        for i, r in enumerate(res):
            r = r.clone()
            r[:, :4] = dt.scale_coords_round((H, W), r[:, :4], (H, W))
            out.append(r.numpy().astype(np.float32))
            if keep_candidates:
                c = dt.candidates(dec[i], conf_thres)
                cands.append((c, dt.nms_rows(c, iou_thres)))
    return (out, cands) if keep_candidates else out


def track(kind, dets_per_frame, **kw):
    """the numpy tracker oracle over a detection stream -> per frame [(id, tlwh, cls, score)] (tracker_np.run)"""
    return tracker_np.run(kind, dets_per_frame, **kw)


def result_rows(stream, min_area=150):
    """track.py:158-172: the rows the CLI would write -> list of (frame (1-based), id, x, y, w, h, cls)"""
    rows = []
    for f, frame in enumerate(stream):
        for tid, tlwh, cls, score in frame:
            if tlwh[2] * tlwh[3] > min_area:
                rows.append((f + 1, int(tid), float(tlwh[0]), float(tlwh[1]), float(tlwh[2]), float(tlwh[3]), float(cls)))
    return rows


def same_detections(a, b, dconf=5e-3, thresholds=(0.15, 0.2, 0.3)):
    """two (n, 6) hand-overs are 'identical' for the tracker: same rows in the same order -- boxes (already integers) and classes equal, confidences
    within `dconf` and on the same side of every threshold ByteTrack compares a confidence with (bytetrack.py:15,69-70,175 at conf_thresh 0.2)"""
    if a.shape != b.shape:
        return False
    if not (np.array_equal(a[:, :4], b[:, :4]) and np.array_equal(a[:, 5], b[:, 5])):
        return False
    if len(a) and np.abs(a[:, 4] - b[:, 4]).max() > dconf:
        return False
    return all(np.array_equal(a[:, 4] > t, b[:, 4] > t) and np.array_equal(a[:, 4] >= t, b[:, 4] >= t) for t in thresholds)


def detection_set_difference(a, b, px=1.0, iou=0.99):
    """rows of a without a partner in b and vice versa -> (only_a, only_b) index lists.  Partner = same class and SURVEY 8a's coordinate bar on the rounded
    boxes: every corner within `px` pixels OR IoU >= `iou` (the several-hundred-pixel boxes of the coarse Detect levels); matched one to one in row order"""
    used = np.zeros(len(b), bool)
    only_a = []
    for i, r in enumerate(a):
        j = np.zeros(0, int)
        if len(b):
            near = np.abs(b[:, :4] - r[:4]).max(1) <= px
            if iou is not None:
                iw = np.clip(np.minimum(b[:, 2], r[2]) - np.maximum(b[:, 0], r[0]), 0, None)
                ih = np.clip(np.minimum(b[:, 3], r[3]) - np.maximum(b[:, 1], r[1]), 0, None)
                ua = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) + (r[2] - r[0]) * (r[3] - r[1]) - iw * ih
                near |= iw * ih >= iou * np.maximum(ua, 1e-9)
            j = np.flatnonzero((~used) & (b[:, 5] == r[5]) & near)
        if len(j):
            used[j[0]] = True
        else:
            only_a.append(i)
    return only_a, np.flatnonzero(~used).tolist()


def grade(root, truth_stream, test_stream, name="chained"):
    """HOTA / CLEAR / Identity of `test_stream` with `truth_stream` as the ground truth, through the product's TrackEval-style harness
    (yolov7-tracker_amd/tracker/trackeval = /root/reference/tracker/trackeval restated; tests/test_trackeval.py pins it to the reference's classes)
    -> {"HOTA": .., "IDF1": .., "MOTA": .., "DetA": .., "AssA": ..} in [0, 1]"""
    from yolov7_tracker_amd.tracker import trackeval
    gt_dir, tr_dir = os.path.join(root, "gt"), os.path.join(root, "trackers", name)
    os.makedirs(gt_dir, exist_ok=True)
    os.makedirs(tr_dir, exist_ok=True)
    T = len(truth_stream)
    with open(os.path.join(gt_dir, "seq.txt"), "w") as f:
        for r in result_rows(truth_stream, min_area=0):
            f.write("%d,%d,%.2f,%.2f,%.2f,%.2f,1,1,1.0\n" % r[:6])
    with open(os.path.join(tr_dir, "seq.txt"), "w") as f:
        for r in result_rows(test_stream, min_area=0):
            f.write("%d,%d,%.2f,%.2f,%.2f,%.2f,1.0,-1,-1,-1\n" % r[:6])
    cfg = trackeval.datasets.MotChallenge2DBox.get_default_dataset_config()
    cfg.update({"GT_FOLDER": gt_dir, "TRACKERS_FOLDER": os.path.join(root, "trackers"), "TRACKERS_TO_EVAL": [name], "SKIP_SPLIT_FOL": True,
                "TRACKER_SUB_FOLDER": "", "SEQ_INFO": {"seq": T}, "GT_LOC_FORMAT": "{gt_folder}/{seq}.txt", "PRINT_CONFIG": False})
    ecfg = trackeval.Evaluator.get_default_eval_config()
    ecfg.update({k: False for k in ecfg if k.startswith("PRINT") or k.startswith("OUTPUT") or k.startswith("PLOT")})
    mcfg = {"METRICS": ["HOTA", "CLEAR", "Identity"], "THRESHOLD": 0.5}
    metrics = [m(mcfg) for m in (trackeval.metrics.HOTA, trackeval.metrics.CLEAR, trackeval.metrics.Identity)]
    res, _ = trackeval.Evaluator(ecfg).evaluate([trackeval.datasets.MotChallenge2DBox(cfg)], metrics)
    r = res["MotChallenge2DBox"][name]["COMBINED_SEQ"]["pedestrian"]
    return {"HOTA": float(np.mean(r["HOTA"]["HOTA"])), "DetA": float(np.mean(r["HOTA"]["DetA"])), "AssA": float(np.mean(r["HOTA"]["AssA"])),
            "IDF1": float(r["Identity"]["IDF1"]), "MOTA": float(r["CLEAR"]["MOTA"]), "IDSW": int(r["CLEAR"]["IDSW"])}
