"""tests/golden/make_golden.py -- regenerates the committed golden vectors.

Runs ONLY in the build container (needs /root/reference): it imports the reference's own Python
sources through oracle/ref_harness.py and records their outputs on seeded inputs, so that the
oracle restatements and the HIP path can be checked against the reference on machines where
/root/reference does not exist (the GPU box).

    python tests/golden/make_golden.py

Writes kalman.npz, tracker_<name>.npz, lap_iou.npz next to this file.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cnative, ref_harness  # noqa: E402
from yolov7_tracker_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rand_state(rng, n, kind):
    """plausible Kalman states: boxes of VisDrone-like size, SPD covariances."""
    mean = np.zeros((n, 8))
    mean[:, 0] = rng.uniform(50, 1200, n)
    mean[:, 1] = rng.uniform(50, 1200, n)
    h = rng.uniform(10, 300, n)
    mean[:, 3] = h
    mean[:, 2] = rng.uniform(10, 200, n) if kind == "botsort" else rng.uniform(0.3, 2.0, n)
    mean[:, 4:] = rng.normal(0, 1.5, (n, 4))
    if kind != "botsort":
        mean[:, 6] = rng.normal(0, 1e-3, n)
    A = rng.normal(0, 1, (n, 8, 8))
    scale = np.array([3, 3, 1e-1, 3, 1, 1, 1e-2, 1.0]) if kind != "botsort" else np.array([3, 3, 3, 3, 1, 1, 1, 1.0])
    cov = np.einsum("nij,nkj->nik", A, A) * 0.05 + np.eye(8) * 2.0
    cov = cov * scale[None, :, None] * scale[None, None, :]
    return mean, cov


def golden_kalman():
    kf = ref_harness.load_tracker().kalman_filter
    classes = {"default": kf.KalmanFilter, "botsort": kf.BoTSORTKalmanFilter, "strongsort": kf.NSAKalmanFilter}
    out = {}
    rng = np.random.default_rng(7)
    n = 24
    for kind, cls in classes.items():
        f = cls()
        mean, cov = rand_state(rng, n, kind)
        z = mean[:, :4] + rng.normal(0, 1.0, (n, 4)) * (np.array([2, 2, 0.02, 2]) if kind != "botsort" else 2.0)
        conf = rng.uniform(0.1, 0.95, n)
        out[kind + "_mean"], out[kind + "_cov"], out[kind + "_z"], out[kind + "_conf"] = mean, cov, z, conf
        mp, cp = f.multi_predict(mean.copy(), cov.copy())
        out[kind + "_pred_mean"], out[kind + "_pred_cov"] = mp, cp
        pm, pc, um, uc, g4, g2 = [], [], [], [], [], []
        for i in range(n):
            if kind == "strongsort":
                a, b = f.project(mean[i], cov[i], conf[i])
                c, d = f.update(mean[i], cov[i], z[i], conf[i])
            else:
                a, b = f.project(mean[i], cov[i])
                c, d = f.update(mean[i], cov[i], z[i])
            pm.append(a); pc.append(b); um.append(c); uc.append(d)
            if hasattr(f, "gating_distance"):  # BoTSORTKalmanFilter has none
                g4.append(f.gating_distance(mean[i], cov[i], z, only_position=False))
                g2.append(f.gating_distance(mean[i], cov[i], z, only_position=True))
        out[kind + "_proj_mean"], out[kind + "_proj_cov"] = np.array(pm), np.array(pc)
        out[kind + "_upd_mean"], out[kind + "_upd_cov"] = np.array(um), np.array(uc)
        out[kind + "_gate4"], out[kind + "_gate2"] = np.array(g4), np.array(g2)
        # initiate: float32 measurement (what STrack.activate passes) and float64 measurement
        z32 = z.astype(np.float32)
        im32, ic32, im64, ic64 = [], [], [], []
        for i in range(n):
            a, b = f.initiate(z32[i]); im32.append(np.asarray(a, dtype=np.float64)); ic32.append(np.asarray(b, dtype=np.float64))
            a, b = f.initiate(z[i]); im64.append(a); ic64.append(b)
        out[kind + "_init32_mean"], out[kind + "_init32_cov"] = np.array(im32), np.array(ic32)
        out[kind + "_init64_mean"], out[kind + "_init64_cov"] = np.array(im64), np.array(ic64)
    out["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(os.path.join(HERE, "kalman.npz"), **out)


TRACKER_CASES = [
    # name, tracker, kalman_format, n_frames, n_obj, seq_idx, drop_every (frames replaced by update_without_detection)
    ("sort_default", "sort", "default", 100, 80, 0, 0),
    ("bytetrack_default", "bytetrack", "default", 100, 80, 0, 0),
    ("bytetrack_default_gaps", "bytetrack", "default", 80, 60, 5, 13),
    ("bytetrack_botsort", "bytetrack", "botsort", 60, 60, 2, 0),
    ("sort_strongsort", "sort", "strongsort", 60, 60, 3, 17),
    ("bytetrack_crowd", "bytetrack", "default", 20, 500, 4, 0),
    # BoT-SORT state path (BASELINE config 3): xywh Kalman + multi_gmc with synthetic 2x3 warps; appearance model off
    ("botsort_gmc", "botsort", "botsort", 80, 80, 6, 11),
    ("botsort_crowd", "botsort", "botsort", 12, 500, 7, 0),
    # DeepSORT association (oracle only so far): matching cascade over appearance + Mahalanobis gate, IoU fallback.
    # The ReID network is replaced at DeepSORT.get_feature by synth.make_features on both sides.
    ("deepsort_default", "deepsort", "default", 80, 60, 8, 0),
    ("deepsort_crowd", "deepsort", "default", 30, 250, 9, 0),
    # other embedding widths: 512 (what OSNet produces; the wave-parallel norms and the tiled distance kernel) and 100 (not a multiple of the
    # kernels' 32-deep chunks: the plain forms)
    ("deepsort_dim512", "deepsort", "default", 60, 70, 10, 0, 512),
    ("deepsort_dim100", "deepsort", "default", 40, 50, 11, 9, 100),
    # appearance features that identify the object (synth.make_identity_features: what a ReID network delivers): a detection is wanted by one
    # track only, the case in which the device solves all cascade levels in one assignment; 20-30 % missed detections keep several ages alive
    ("deepsort_identity", "deepsort", "default", 60, 60, 20, 0, 128, "identity", 0.2),
    ("deepsort_identity512", "deepsort", "default", 50, 80, 21, 0, 512, "identity", 0.3),
]


def pack_tracks(frames):
    """list of per-frame [(id, tlwh, cls, score)] -> flat arrays."""
    fr, ids, tlwh, cls, score = [], [], [], [], []
    for f, rows in enumerate(frames):
        for r in rows:
            fr.append(f); ids.append(r[0]); tlwh.append(r[1]); cls.append(r[2]); score.append(r[3])
    return (np.array(fr, np.int32), np.array(ids, np.int32), np.array(tlwh, np.float64).reshape(-1, 4), np.array(cls, np.float32),
            np.array(score, np.float32))


def golden_tracker(only=None):
    for case in TRACKER_CASES:
        name, trk, fmt, nf, nobj, seq, drop = case[:7]
        feat_dim = case[7] if len(case) > 7 else 128
        if only and name not in only:
            continue
        identity = len(case) > 8 and case[8] == "identity"
        feat_fn = (lambda b, _d=feat_dim: synth.make_features(b, dim=_d))
        if identity:
            dets, feat_fn = synth.make_identity_features(nf, nobj, 1280, seq_idx=seq, dim=feat_dim, miss=case[9])
        else:
            dets = synth.make_detections(nf, nobj, seq_idx=seq)
        if drop:
            dets = [None if (i % drop == drop - 1) else d for i, d in enumerate(dets)]
        warps = synth.make_warps(nf, seq_idx=seq) if trk == "botsort" else None
        ref = ref_harness.run_reference_tracker(trk, dets, opts=ref_harness.make_opts(kalman_format=fmt), warps=warps,
                                                feature_fn=feat_fn if trk == "deepsort" else None)
        fr, ids, tlwh, cls, score = pack_tracks(ref)
        counts = np.array([-1 if d is None else len(d) for d in dets], np.int32)
        flat = np.concatenate([d for d in dets if d is not None], 0).astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "tracker_%s.npz" % name), tracker=np.array(trk), kalman_format=np.array(fmt),
                            det_counts=counts, dets=flat, frame=fr, track_id=ids, tlwh=tlwh, cls=cls, score=score,
                            warps=np.zeros((0, 2, 3)) if warps is None else warps,
                            numpy_version=np.array(np.__version__), feat_dim=np.array(feat_dim),
                            feat_kind=np.array("identity" if identity else "size"), scene=np.array([nf, nobj, seq], np.int64),
                            feat_miss=np.array(case[9] if identity else 0.0))
        print(name, "rows", len(ids), "max id", ids.max())


def golden_lap_iou():
    """Third-party kernels (lap.lapjv / cython_bbox.bbox_overlaps): NOT pinned by the reference (packages absent).
    These vectors come from the C restatement (oracle/y7t_oracle.c) cross-checked with scipy here."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(11)
    out = {}
    k = 0
    for nr, nc, lim in [(1, 1, 0.9), (5, 3, 0.9), (3, 7, 0.5), (20, 20, 0.7), (37, 50, 0.9), (64, 40, 0.5), (90, 110, 0.9)]:
        a = np.concatenate([rng.uniform(0, 1000, (nr, 2)), np.zeros((nr, 2))], 1)
        a[:, 2:] = a[:, :2] + rng.uniform(10, 120, (nr, 2))
        b = a[rng.integers(0, nr, nc)] + rng.normal(0, 6, (nc, 4))
        b = np.round(b)
        iou = cnative.bbox_overlaps(a, b)
        cost = 1 - iou
        opt, x, y = cnative.lapjv(cost, extend_cost=True, cost_limit=lim)
        # cross-check optimality with scipy on the explicit extended matrix
        n = nr + nc
        ext = np.full((n, n), lim / 2.0); ext[nr:, nc:] = 0; ext[:nr, :nc] = cost
        r, c = linear_sum_assignment(ext)
        tot_ref = ext[r, c].sum()
        xe = np.where(x >= 0, x, -1)
        tot = sum(cost[i, xe[i]] for i in range(nr) if xe[i] >= 0) + (lim / 2.0) * ((x < 0).sum() + (y < 0).sum())
        assert abs(tot - tot_ref) < 1e-9, (tot, tot_ref)
        out["a%d" % k], out["b%d" % k], out["iou%d" % k], out["lim%d" % k] = a, b, iou, np.array(lim)
        out["x%d" % k], out["y%d" % k] = x.astype(np.int32), y.astype(np.int32)
        k += 1
    out["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "lap_iou.npz"), **out)


def golden_detector():
    """reference models.yolo.Model (built from ITS yaml) loaded with this repo's seeded random state dict, run on a
    seeded input: decoded head + NMS output.  Pins oracle/detector_torch.py where /root/reference is absent."""
    import torch
    from oracle import detector_torch as dt
    from yolov7_tracker_amd.detector import arch, graph, weights
    out = {}
    for name, cfg, nc, hw in (("tiny", "cfg/deploy/yolov7-tiny.yaml", 80, (64, 96)), ("w6", "cfg/deploy/yolov7-w6.yaml", 10, (128, 128))):
        spec = arch.ARCHS["yolov7-" + name](nc)
        nodes, _ = graph.parse(spec)
        plan = graph.lower(graph.parse(spec)[0], hw[0], hw[1], 1)
        sd = weights.calibrate_bn(nodes, weights.random_state_dict(plan.wlayout, 0), hw=(128, 128), seed=0)
        m = ref_harness.build_reference_model(cfg, nc)
        missing = m.load_state_dict(sd, strict=False)
        assert not [k for k in missing.unexpected_keys], missing.unexpected_keys
        assert all(("anchor" in k or "num_batches_tracked" in k) for k in missing.missing_keys), missing.missing_keys
        img = torch.rand((1, 3) + hw, generator=torch.Generator().manual_seed(5))
        with torch.no_grad():
            dec = m(img)[0]
        det = ref_harness.load_detector()
        det.general.torchvision.ops.nms = lambda b, s, t: torch.from_numpy(cnative.nms(b.numpy(), s.numpy(), t))
        pred = dec.clone()
        pred[..., 4] = torch.rand(pred.shape[:2], generator=torch.Generator().manual_seed(6)) * 0.3
        nms = det.general.non_max_suppression(pred.clone(), conf_thres=0.01)[0]
        out[name + "_img"], out[name + "_decoded"], out[name + "_pred"], out[name + "_nms"] = img.numpy(), dec.numpy(), pred.numpy(), nms.numpy()
        out[name + "_sd_checksum"] = np.array(sum(float(v.double().sum()) for v in sd.values()))
    np.savez_compressed(os.path.join(HERE, "detector.npz"), **out)


def golden_trackeval():
    """the REFERENCE's TrackEval classes (MotChallenge2DBox pre-processing, HOTA / CLEAR / Identity) on the scenario of
    tests/trackeval_case.py -> per-sequence and combined metric fields.  Pins yolov7_tracker_amd.tracker.trackeval."""
    import json
    import tempfile
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from tests import trackeval_case
    from oracle import ref_trackeval
    ns = ref_trackeval.load()
    with tempfile.TemporaryDirectory() as root:
        cfg = trackeval_case.build(root)
        ds = ns.MotChallenge2DBox({**ns.MotChallenge2DBox.get_default_dataset_config(), **cfg})
        metrics = [ns.HOTA(), ns.CLEAR({"THRESHOLD": 0.5, "PRINT_CONFIG": False}), ns.Identity({"THRESHOLD": 0.5, "PRINT_CONFIG": False})]
        out, per_seq = {}, {m.get_name(): {} for m in metrics}
        for seq in sorted(cfg["SEQ_INFO"]):
            raw = ds.get_raw_seq_data("bytetrack_oracle", seq)
            data = ds.get_preprocessed_seq_data(raw, "pedestrian")
            res = {m.get_name(): m.eval_sequence(data) for m in metrics}
            for m in metrics:
                per_seq[m.get_name()][seq] = res[m.get_name()]
            out[seq] = trackeval_case.flatten(res)
            out[seq]["counts"] = [data["num_gt_dets"], data["num_tracker_dets"], data["num_gt_ids"], data["num_tracker_ids"]]
        out["COMBINED_SEQ"] = trackeval_case.flatten({m.get_name(): m.combine_sequences(per_seq[m.get_name()]) for m in metrics})
    with open(os.path.join(HERE, "trackeval_metrics.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    assert ref_harness.available(), "needs /root/reference"
    if len(sys.argv) > 2 and sys.argv[1] == "--tracker":      # python make_golden.py --tracker name[,name...]: (re)record only these sequences
        golden_tracker(only=sys.argv[2].split(","))
        sys.exit(0)
    golden_kalman()
    golden_tracker()
    golden_lap_iou()
    golden_detector()
    golden_trackeval()
    print("golden vectors written to", HERE)
