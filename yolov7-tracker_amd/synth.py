"""Deterministic synthetic workloads for the tracking hot path.

No dataset or trained checkpoint ships with the reference, so every parity test and
benchmark in this repo runs on seeded synthetic sequences:

* `make_detections` -- ground-truth rectangles moving with constant velocity + jitter,
  emitted at the tracker's `(N, 6)` seam `[x1, y1, x2, y2, conf, cls]` exactly as
  `post_process_v7` hands them over (integer-rounded corners, float32;
  /root/reference/tracker/track.py:234-244).
* `make_frames` -- uint8 BGR frames of VisDrone-like shape for the detector.
"""
import numpy as np

BASE_SEED = 20260924


def _tracks(rng, n_obj, size):
    w = np.clip(rng.lognormal(np.log(30.0), 0.5, n_obj), 8, 300)
    h = np.clip(rng.lognormal(np.log(30.0), 0.5, n_obj) * 1.6, 8, 300)
    cx = rng.uniform(0.05 * size, 0.95 * size, n_obj)
    cy = rng.uniform(0.05 * size, 0.95 * size, n_obj)
    vx = rng.normal(0, 2.0, n_obj)
    vy = rng.normal(0, 2.0, n_obj)
    cls = rng.integers(0, 10, n_obj)
    conf0 = rng.uniform(0.1, 0.95, n_obj)
    return cx, cy, w, h, vx, vy, cls, conf0


_FEAT_DIM = 128


def make_detections(n_frames=100, n_obj=80, size=1280, seq_idx=0, miss=0.10, fp=0.05, conf_jitter=0.05, ground_truth=None, bounce=False):
    """-> list of float32 (N_t, 6) arrays, one per frame.  `ground_truth`: a list that receives, per frame, the float64
    (n, 7) rows `[id (1-based), x, y, w, h, cls, detected]` of the true rectangles (clipped to the image; objects that left it
    are dropped) -- what make_ground_truth returns.  bounce=True: objects are reflected at the image border instead of leaving
    (long sequences keep ~n_obj objects per frame: bench.py; the default keeps the generator the golden sequences were recorded
    with: there the constant-velocity objects drift out, ~27 of 80 left after 800 frames)."""
    rng = np.random.default_rng(BASE_SEED + seq_idx)
    cx, cy, w, h, vx, vy, cls, conf0 = _tracks(rng, n_obj, size)
    out = []
    for _ in range(n_frames):
        cx = cx + vx
        cy = cy + vy
        if bounce:
            for c, v in ((cx, vx), (cy, vy)):
                lo, hi = c < 0.02 * size, c > 0.98 * size
                v[lo] = np.abs(v[lo])
                v[hi] = -np.abs(v[hi])
        if ground_truth is not None:
            gx1, gy1 = np.clip(cx - w / 2, 0, size), np.clip(cy - h / 2, 0, size)
            gx2, gy2 = np.clip(cx + w / 2, 0, size), np.clip(cy + h / 2, 0, size)
            vis = (gx2 - gx1 >= 2) & (gy2 - gy1 >= 2)
            gt_rows = np.stack([np.arange(1, n_obj + 1, dtype=np.float64), gx1, gy1, gx2 - gx1, gy2 - gy1, cls.astype(np.float64),
                                np.zeros(n_obj)], 1)
        jx = rng.normal(0, 0.5, n_obj)
        jy = rng.normal(0, 0.5, n_obj)
        keep = rng.random(n_obj) >= miss
        x1 = cx + jx - w / 2
        y1 = cy + jy - h / 2
        x2 = x1 + w
        y2 = y1 + h
        conf = np.clip(conf0 + rng.normal(0, conf_jitter, n_obj), 0.02, 0.99)
        rows = np.stack([x1, y1, x2, y2, conf, cls.astype(np.float64)], 1)[keep]
        if ground_truth is not None:
            gt_rows[:, 6] = keep
            ground_truth.append(gt_rows[vis])
        n_fp = rng.binomial(n_obj, fp)
        if n_fp:
            fw = np.clip(rng.lognormal(np.log(30.0), 0.5, n_fp), 8, 300)
            fh = np.clip(rng.lognormal(np.log(30.0), 0.5, n_fp) * 1.6, 8, 300)
            fx = rng.uniform(0, size - 8, n_fp)
            fy = rng.uniform(0, size - 8, n_fp)
            fr = np.stack([fx, fy, fx + fw, fy + fh, rng.uniform(0.1, 0.95, n_fp),
                           rng.integers(0, 10, n_fp).astype(np.float64)], 1)
            rows = np.concatenate([rows, fr], 0)
        rows[:, :4] = np.clip(np.round(rows[:, :4]), 0, size)
        ok = (rows[:, 2] - rows[:, 0] >= 2) & (rows[:, 3] - rows[:, 1] >= 2)
        rows = rows[ok]
        # reject exact duplicate boxes so assignment optima stay unique
        _, first = np.unique(rows[:, :4], axis=0, return_index=True)
        rows = rows[np.sort(first)]
        # the detector hands rows over sorted by confidence (NMS output order)
        rows = rows[np.argsort(-rows[:, 4], kind="stable")]
        out.append(rows.astype(np.float32))
    return out



def make_features(boxes, seed=0, dim=_FEAT_DIM):
    """Deterministic stand-in for a ReID embedding at DeepSORT's `get_feature(tlbrs, ori_img)` seam (tracker/deepsort.py:19-41; no
    ReID checkpoint ships with the reference, weights/ckpt.t7): a fixed random projection of what stays constant for an object in
    these scenes -- its box width and height -- through a few non-linear terms to a unit vector of dimension `dim` (128), float32.
    boxes: (N, >=4) [x1, y1, x2, y2, ...]."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(len(boxes), -1)
    rng = np.random.default_rng(BASE_SEED + 1000 + seed)
    proj = rng.normal(0, 1, (6, dim)).astype(np.float32)
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    code = np.stack([w / 32.0, h / 32.0, np.sqrt(np.maximum(w * h, 0)) / 32.0, w / np.maximum(h, 1.0), np.sin(w / 7.0), np.cos(h / 9.0)], 1).astype(np.float32)
    f = code @ proj
    return (f / np.maximum(np.linalg.norm(f, axis=1, keepdims=True), 1e-12)).astype(np.float32)


def make_identity_features(n_frames=60, n_obj=60, size=1280, seq_idx=0, dim=128, noise=0.15, **kw):
    """A scene whose appearance features IDENTIFY the object (what a ReID network delivers): -> (dets, feature_fn).  Every detection of object k
    gets unit(base_k + noise * N(0, 1)), false positives a random unit vector; feature_fn(boxes) looks the vectors up by the box (the seam
    DeepSORT.get_feature(tlbrs, ori_img) only sees boxes).  With such features a detection is an appearance candidate of ONE track, which is
    when the matching cascade's levels do not compete for detections."""
    gt = []
    dets = make_detections(n_frames, n_obj, size, seq_idx, ground_truth=gt, **kw)
    rng = np.random.default_rng(BASE_SEED + 2000 + seq_idx)
    base = rng.normal(0, 1, (n_obj + 1, dim)).astype(np.float32)
    table = {}
    for d, g in zip(dets, gt):
        f = rng.normal(0, 1, (len(d), dim)).astype(np.float32)
        if len(g) and len(d):
            gc = np.stack([g[:, 1] + g[:, 3] / 2, g[:, 2] + g[:, 4] / 2], 1)
            dc = np.stack([(d[:, 0] + d[:, 2]) / 2, (d[:, 1] + d[:, 3]) / 2], 1)
            dist = np.abs(dc[:, None, :] - gc[None, :, :]).max(2)
            j = dist.argmin(1)
            hit = dist[np.arange(len(d)), j] < 4.0
            f[hit] = base[g[j[hit], 0].astype(int)] + np.float32(noise) * f[hit]
        f = (f / np.maximum(np.linalg.norm(f, axis=1, keepdims=True), 1e-12)).astype(np.float32)
        for row, v in zip(d, f):
            table[tuple(float(x) for x in row[:4])] = v

    def feature_fn(boxes):
        boxes = np.asarray(boxes, dtype=np.float32).reshape(len(boxes), -1)
        return np.stack([table[tuple(float(x) for x in b[:4])] for b in boxes]) if len(boxes) else np.zeros((0, dim), np.float32)
    return dets, feature_fn


def make_ground_truth(n_frames=100, n_obj=80, size=1280, seq_idx=0, **kw):
    """ground truth of the sequence make_detections(...) observes: per frame float64 (n, 7) `[id, x, y, w, h, cls, detected]`"""
    gt = []
    make_detections(n_frames, n_obj, size, seq_idx, ground_truth=gt, **kw)
    return gt


def write_mot_gt(path, gt, single_class=True):
    """MOTChallenge gt.txt: frame,id,x,y,w,h,conf(1 = evaluate),class,visibility (class 1 = pedestrian when single_class)"""
    with open(path, "w") as f:
        for t, rows in enumerate(gt):
            for r in rows:
                f.write("%d,%d,%.2f,%.2f,%.2f,%.2f,1,%d,1.0\n" % (t + 1, int(r[0]), r[1], r[2], r[3], r[4], 1 if single_class else int(r[5]) + 1))


def make_frames(n_frames=4, n_obj=80, size=1280, seq_idx=0):
    """-> uint8 (n_frames, size, size, 3) BGR frames: smooth background + moving filled rectangles."""
    rng = np.random.default_rng(BASE_SEED + 1000 + seq_idx)
    cx, cy, w, h, vx, vy, cls, _ = _tracks(rng, n_obj, size)
    colors = rng.integers(0, 256, (n_obj, 3))
    coarse = rng.integers(60, 200, (40, 40, 3)).astype(np.float32)
    rep = size // 40 + 1
    bg = np.kron(coarse, np.ones((rep, rep, 1), np.float32))[:size, :size]
    # cheap separable box blur to make it low-frequency
    k = max(rep // 2, 1)
    bg = (bg + np.roll(bg, k, 0) + np.roll(bg, -k, 0) + np.roll(bg, k, 1) + np.roll(bg, -k, 1)) / 5.0
    frames = np.empty((n_frames, size, size, 3), np.uint8)
    for t in range(n_frames):
        cx = cx + vx
        cy = cy + vy
        img = bg.copy()
        for i in range(n_obj):
            x1 = int(np.clip(cx[i] - w[i] / 2, 0, size - 1))
            y1 = int(np.clip(cy[i] - h[i] / 2, 0, size - 1))
            x2 = int(np.clip(cx[i] + w[i] / 2, x1 + 1, size))
            y2 = int(np.clip(cy[i] + h[i] / 2, y1 + 1, size))
            img[y1:y2, x1:x2] = colors[i]
        frames[t] = img.astype(np.uint8)
    return frames


def make_warps(n_frames=100, seq_idx=0, rot=0.002, shift=2.0):
    """-> (n_frames, 2, 3) float64 camera-motion matrices near the identity (what BoT-SORT's GMC.apply would estimate from
    ORB matches, /root/reference/tracker/botsort.py:13-248 -- the estimation itself is OpenCV and out of scope)."""
    rng = np.random.default_rng(BASE_SEED + 5000 + seq_idx)
    H = np.zeros((n_frames, 2, 3))
    H[:, :2, :2] = np.eye(2) + rng.normal(0, rot, (n_frames, 2, 2))
    H[:, :, 2] = rng.normal(0, shift, (n_frames, 2))
    return H
