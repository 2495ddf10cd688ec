"""GPU, end to end through the reference's CLI surface (tracker/track.py flags, result-file format): BASELINE config 1
(YOLOv7-tiny 640x640 + SORT, one synthetic 100-frame sequence) and a short YOLOv7-w6 + ByteTrack run.  The detector runs
on every frame (random weights), the tracker consumes the scene's detections (--synthetic_dets); the written result file must
be byte-identical to what the CPU oracle tracker produces on the same detections."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_file(kind, n_frames, n_obj, size, fmt="default", min_area=150):
    from oracle import tracker_np
    from yolov7_tracker_amd import synth
    dets = synth.make_detections(n_frames, n_obj, size, 0)
    out = tracker_np.run(kind, dets, kalman_format=fmt)
    lines = []
    for f, rows in enumerate(out):
        for tid, tlwh, cls, score in rows:
            if tlwh[2] * tlwh[3] > min_area:
                lines.append(f'{f + 1},{tid},{tlwh[0]:.2f},{tlwh[1]:.2f},{tlwh[2]:.2f},{tlwh[3]:.2f},1.0,-1,-1,-1\n')
    return "".join(lines)


@pytest.mark.parametrize("tracker,model,size,frames,objs,batch", [("sort", "random:yolov7-tiny", 640, 100, 40, 1), ("bytetrack", "random:yolov7-w6", 1280, 12, 80, 1),
                                                                  ("bytetrack", "random:yolov7-tiny", 640, 50, 40, 8)])
def test_track_cli_synthetic(tmp_path, tracker, model, size, frames, objs, batch):
    from yolov7_tracker_amd.tracker import track
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    BaseTrack._count = 0
    folder = track.cli(["--dataset", "synthetic", "--tracker", tracker, "--model_path", model, "--nc", "10", "--img_size", str(size),
                        "--synthetic_dets", "--synthetic_frames", str(frames), "--synthetic_objs", str(objs), "--results_root", str(tmp_path),
                        "--batch", str(batch)])
    path = os.path.join(folder, "synthetic-000.txt")
    got = open(path).read()
    want = oracle_file(tracker, frames, objs, size)
    assert len(got) > 0
    # ids and the %.2f-rounded boxes identical (a box coordinate within 1e-6 of a rounding boundary may differ in the last digit)
    gl, wl = got.splitlines(), want.splitlines()
    assert len(gl) == len(wl)
    diff = [i for i, (a, b) in enumerate(zip(gl, wl)) if a != b]
    for i in diff:
        a, b = gl[i].split(","), wl[i].split(",")
        assert a[:2] == b[:2] and np.allclose([float(v) for v in a[2:6]], [float(v) for v in b[2:6]], atol=0.011)
    assert len(diff) <= max(2, len(gl) // 500)
    # --track_eval (default on): ground truth written next to the results, HOTA / CLEAR / Identity summary next to the result file
    summary = open(os.path.join(folder, "pedestrian_summary.txt")).read().split("\n")
    vals = dict(zip(summary[0].split(), summary[1].split()))
    assert float(vals["MOTA"]) > 50 and float(vals["IDF1"]) > 50 and float(vals["HOTA"]) > 40


def test_track_cli_deepsort_with_device_reid(tmp_path):
    """BASELINE config 4 through the CLI: --tracker deepsort with the OSNet x0_25 embedding network on the device (seeded random weights:
    --reid_model_path random:osnet), crops taken from the synthetic frames.  The oracle DeepSORT fed with the SAME embeddings (computed by the
    same extractor from the same frames) must write the same file: ids and boxes of every row"""
    from oracle import tracker_np
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.tracker import track
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.reid import ReIDExtractor
    frames_n, objs, size = 30, 40, 640
    BaseTrack._count = 0
    folder = track.cli(["--dataset", "synthetic", "--tracker", "deepsort", "--reid_model_path", "random:osnet", "--model_path", "random:yolov7-tiny", "--nc", "10",
                        "--img_size", str(size), "--synthetic_dets", "--synthetic_frames", str(frames_n), "--synthetic_objs", str(objs),
                        "--results_root", str(tmp_path)])
    got = open(os.path.join(folder, "synthetic-000.txt")).read().splitlines()
    frames = synth.make_frames(frames_n, objs, size, 0)
    dets = synth.make_detections(frames_n, objs, size, 0)
    ext, state = ReIDExtractor(None, arch="osnet", max_crops=512), {"f": 0}

    def feature_fn(tlbrs):          # called once per frame, in order, with the rows above the confidence threshold (deepsort.py:98)
        f = state["f"]
        state["f"] += 1
        return ext.features_for_boxes(frames[f], tlbrs).cpu().numpy()
    want = []
    for f, rows in enumerate(tracker_np.run("deepsort", dets, feature_fn=feature_fn)):
        for tid, tlwh, cls, score in rows:
            if tlwh[2] * tlwh[3] > 150:
                want.append(f'{f + 1},{tid},{tlwh[0]:.2f},{tlwh[1]:.2f},{tlwh[2]:.2f},{tlwh[3]:.2f},1.0,-1,-1,-1')
    assert len(got) > 0.5 * sum(len(d) for d in dets) and len(got) == len(want)
    same = sum(a == b for a, b in zip(got, want))
    assert same >= 0.99 * len(want), (same, len(want))
    assert [l.split(",")[:2] for l in got] == [l.split(",")[:2] for l in want]          # frame, id of every row


def _rows(txt):
    return [tuple(l.split(",")[:2]) for l in txt.splitlines()]


@pytest.mark.parametrize("shape,resampled", [((360, 640), False), ((540, 960), True)])
def test_track_cli_image_folder_device_preprocess(tmp_path, shape, resampled):
    """A folder of non-square frames on disk (the reference's 'origin' data format), the detector's REAL decode+NMS output feeding the
    tracker: the loader letterboxes on the host (tracker_dataloader.py:100-130), with --device_preprocess the raw frame is letterboxed on
    the GPU.  360x640 frames need padding only (both paths produce the same pixels): the two result files must be identical.  540x960
    frames are resampled (the device filter differs from the host's in <1 % of the pixels by one grey level, test_detector_gpu): the
    tracks must still agree -- same (frame, id) rows on >= 95 % of the rows."""
    from PIL import Image
    from yolov7_tracker_amd.tracker import track, tracker_dataloader
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from oracle import letterbox_np as lb
    seq = tmp_path / "data" / "seqs" / "uav0001"
    seq.mkdir(parents=True)
    rng = np.random.default_rng(0)
    small = rng.integers(0, 256, (shape[0] // 16 + 4, shape[1] // 16 + 4, 3)).astype(np.float32)
    big = np.kron(small, np.ones((16, 16, 1), np.float32)).astype(np.uint8)
    for i in range(4):                                     # a slowly panning scene
        Image.fromarray(big[2 * i:2 * i + shape[0], 3 * i:3 * i + shape[1]]).save(seq / ("%07d.png" % (i + 1)))
    cfgs = {'DATASET_ROOT': str(tmp_path / "data"), 'SEQ_SUBDIR': 'seqs', 'CERTAIN_SEQS': [None], 'IGNORE_SEQS': [None],
            'CATEGORY_DICT': {}, 'YAML_DICT': ''}
    outs = []
    for extra in ([], ["--device_preprocess"]):
        BaseTrack._count = 0
        opts = track.build_parser().parse_args(["--dataset", "visdrone", "--tracker", "sort", "--model_path", "random:yolov7-tiny", "--nc", "10",
                                                "--img_size", "640", "--results_root", str(tmp_path / ("res%d" % len(outs)))] + extra)
        folder = track.main(opts, cfgs)
        outs.append(open(os.path.join(folder, "uav0001.txt")).read())
    loader = tracker_dataloader.TrackerLoader(str(seq), 640, model_stride=32)
    img, ori = loader[0]
    ref = lb.letterbox(ori.numpy(), new_shape=(640, 640), stride=32)
    assert tuple(img.shape[1:]) == ref.shape[:2] == (384, 640)          # auto=True: padded to the stride multiple only
    host, dev = _rows(outs[0]), _rows(outs[1])
    assert len(host) >= 20 and len(dev) >= 20, (len(host), len(dev))       # the tracker really was fed detections
    if not resampled:
        assert outs[0] == outs[1]
    else:
        common = len(set(host) & set(dev))
        print("device-preprocess vs host-loader: %d / %d rows with the same (frame, id)" % (common, len(host)))
        assert common >= 0.95 * max(len(host), len(dev))
