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


@pytest.mark.parametrize("tracker,model,size,frames,objs", [("sort", "random:yolov7-tiny", 640, 100, 40), ("bytetrack", "random:yolov7-w6", 1280, 12, 80)])
def test_track_cli_synthetic(tmp_path, tracker, model, size, frames, objs):
    from yolov7_tracker_amd.tracker import track
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    BaseTrack._count = 0
    folder = track.cli(["--dataset", "synthetic", "--tracker", tracker, "--model_path", model, "--nc", "10", "--img_size", str(size),
                        "--synthetic_dets", "--synthetic_frames", str(frames), "--synthetic_objs", str(objs), "--results_root", str(tmp_path)])
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
