"""CPU: the dataset side of the CLI drop-in -- sequence listing and frame loading for both `--data_format`s of the reference
(/root/reference/tracker/track.py:93-109,126; tracker/tracker_dataloader.py:21-62,64-98)."""
import os
import types

import numpy as np
import pytest
import torch


def _dataset(tmp_path, seqs=("uav0000009_03358_v", "uav0000073_00600_v", "uav0000120_04775_v"), n=3, hw=(48, 80)):
    from PIL import Image
    root = tmp_path / "datasets"
    rng = np.random.default_rng(7)
    lines = []
    for si, s in enumerate(seqs):
        d = root / "VisDrone" / "images" / s
        d.mkdir(parents=True)
        for i in range(n + si):
            Image.fromarray(rng.integers(0, 255, hw + (3,), dtype=np.uint8)).save(str(d / ("%07d.png" % (i + 1))))
            lines.append("VisDrone/images/%s/%07d.png" % (s, i + 1))
    return root, lines


def test_yolo_data_format_sequence_list_and_loader(tmp_path):
    """`--data_format yolo`: ./<dataset>/test.txt lists image paths relative to the data root; the sequences are the parent-folder names (track.py:95-101), every
    sequence's loader filters ITS lines out of that file (tracker_dataloader.py:44-53).  Same frames, same tensors as the 'origin' loader pointed at the folder."""
    from yolov7_tracker_amd.tracker import track, tracker_dataloader
    root, lines = _dataset(tmp_path)
    work = tmp_path / "work"
    (work / "visdrone").mkdir(parents=True)
    rng = np.random.default_rng(0)
    (work / "visdrone" / "test.txt").write_text("\n".join(lines[i] for i in rng.permutation(len(lines))[::-1]) + "\n")      # any order: the list is sorted
    cfgs = {"DATASET_ROOT": str(root), "CERTAIN_SEQS": [None], "IGNORE_SEQS": ["uav0000073_00600_v"]}
    opts = types.SimpleNamespace(data_format="yolo", dataset="visdrone", yolo_root=str(work))
    seqs, data_root = track.sequence_list(opts, cfgs)
    assert seqs == ["uav0000009_03358_v", "uav0000120_04775_v"] and data_root is None
    cfgs2 = dict(cfgs, CERTAIN_SEQS=["uav0000120_04775_v"])
    assert track.sequence_list(opts, cfgs2)[0] == ["uav0000120_04775_v"]
    path = os.path.join(str(work), "visdrone", "test.txt")
    for s, n in (("uav0000009_03358_v", 3), ("uav0000120_04775_v", 5)):
        ly = tracker_dataloader.TrackerLoader(path, 64, "yolo", s, model_stride=32, yolo_data_root=str(root))
        lo = tracker_dataloader.TrackerLoader(str(root / "VisDrone" / "images" / s), 64, "origin", s, model_stride=32)
        assert len(ly) == len(lo) == n
        got = sorted(ly.img_files)
        assert got == [str(root / "VisDrone" / "images" / s / f) for f in sorted(lo.img_files)]
        ly.img_files = got                                                   # (the reference keeps file order; compare frame by frame in sorted order)
        for i in range(n):
            a, a0 = ly[i]
            b, b0 = lo[i]
            assert torch.equal(a, b) and torch.equal(a0, b0) and a.shape[0] == 3 and a.dtype == torch.float32
    with pytest.raises(NotImplementedError):
        track.sequence_list(types.SimpleNamespace(data_format="coco", dataset="visdrone", yolo_root=str(work)), cfgs)
    with pytest.raises(NotImplementedError):
        tracker_dataloader.TrackerLoader(path, 64, "coco", "x")


def test_origin_sequence_list(tmp_path):
    from yolov7_tracker_amd.tracker import track
    root, _ = _dataset(tmp_path)
    cfgs = {"DATASET_ROOT": str(root), "SEQ_SUBDIR": "VisDrone/images", "CERTAIN_SEQS": [None], "IGNORE_SEQS": []}
    seqs, data_root = track.sequence_list(types.SimpleNamespace(data_format="origin", dataset="visdrone"), cfgs)
    assert seqs == sorted(os.listdir(data_root)) and len(seqs) == 3
