"""GPU: the device ReID path (csrc/y7t_reid.hip, tracker/reid.py) against oracle/reid_torch.py (pinned to the reference's OSNet class in
tests/test_reid_oracle.py), and DeepSORT end to end with it (BASELINE config 4: appearance features from OSNet x0_25 crops)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def extractor():
    from yolov7_tracker_amd.tracker.reid import ReIDExtractor
    return ReIDExtractor(None, seed=3, max_crops=256, fused=False)      # the fp32 op list: the exact path


@pytest.fixture(scope="module")
def fused_extractor():
    from yolov7_tracker_amd.tracker.reid import ReIDExtractor
    e = ReIDExtractor(None, seed=3, max_crops=1024)                      # default for x0_25 / 128 x 64: the one-kernel MFMA path
    assert e.fused
    return e


def test_osnet_forward_matches_oracle(extractor):
    from oracle import reid_torch
    x = torch.randn((37, 3, 128, 64), generator=torch.Generator().manual_seed(1))
    got = extractor.forward_crops(x.permute(0, 2, 3, 1).contiguous()).cpu()
    want = reid_torch.osnet_forward(extractor.sd, x)
    assert got.shape == want.shape == (37, 512) and float(want.abs().mean()) > 0.05
    # fp32 on both sides; only the summation order differs (BatchNorm folded into the weights on the device side)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * float(want.abs().max()))


def test_crop_resize_normalise_matches_oracle(extractor):
    from oracle import reid_torch
    from yolov7_tracker_amd import synth
    frame = synth.make_frames(1, 80, 640, seq_idx=3)[0]
    boxes = np.array([[10, 20, 60, 150], [100.7, 0.2, 130.9, 64.5], [300, 300, 364, 428], [600, 500, 640, 640], [5, 5, 13, 21], [200, 100, 520, 600]], np.float32)
    want_x = reid_torch.preprocess(frame, boxes)
    want = reid_torch.osnet_forward(extractor.sd, want_x)
    got = extractor.features_for_boxes(frame, boxes).cpu()
    # the device crop (first arena buffer) against the oracle's preprocessing
    crop = extractor._arena[:6 * 128 * 64 * 3].view(6, 128, 64, 3).cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(crop.numpy(), want_x.numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-3, atol=1e-3 * float(want.abs().max()))
    # host-crop entry point (Extractor.__call__ semantics) gives the same features
    crops = [frame[int(b[1]):int(b[3]), int(b[0]):int(b[2])] for b in boxes]
    np.testing.assert_allclose(extractor(crops), got.numpy(), rtol=1e-4, atol=1e-4 * float(want.abs().max()))


def test_fused_osnet_kernel_matches_oracle(fused_extractor, extractor):
    """csrc/y7t_reid_fused.hip (one workgroup per crop, fp16 storage / fp32 accumulate on the MFMA units) against the fp32 oracle from the FRAME:
    crop + resize + normalise + ~70 layers.  Tolerance: every stored activation is rounded to fp16 (rel. 2^-11 = 4.9e-4) and the errors add
    incoherently over the ~35 layers of the longest path -> a few 1e-3 of the feature scale; measured 5e-4 of max|feature|, cosine 1 - 1e-7."""
    from oracle import reid_torch
    from yolov7_tracker_amd import synth
    frame = synth.make_frames(1, 80, 640, seq_idx=3)[0]
    rng = np.random.default_rng(5)
    xy, wh = rng.uniform(0, 560, (60, 2)), rng.uniform(8, 200, (60, 2))
    boxes = np.concatenate([xy, np.minimum(xy + wh, 700)], 1).astype(np.float32)      # some reach past the right / bottom edge (numpy clips the slice)
    boxes[:6] = [[10, 20, 60, 150], [100.7, 0.2, 130.9, 64.5], [300, 300, 364, 428], [600, 500, 640, 640], [5, 5, 13, 21], [200, 100, 520, 600]]
    want = reid_torch.osnet_forward(fused_extractor.sd, reid_torch.preprocess(frame, boxes)).numpy()
    got = fused_extractor.features_for_boxes(frame, boxes).cpu().numpy()
    assert np.isfinite(got).all() and float(np.abs(want).mean()) > 0.05
    scale = float(np.abs(want).max())
    assert np.abs(got - want).max() <= 3e-3 * scale, np.abs(got - want).max() / scale
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert cos.min() >= 1 - 1e-5, cos.min()
    # and against the device fp32 path on the same crops, including an empty box (zero crop on both)
    boxes[7] = [50, 50, 50, 90]
    a, b = extractor.features_for_boxes(frame, boxes).cpu().numpy(), fused_extractor.features_for_boxes(frame, boxes).cpu().numpy()
    assert np.abs(a - b).max() <= 3e-3 * scale


def test_fused_batch_over_frames_equals_per_frame(fused_extractor):
    """y7t_reid_forward_batch: crops of several frames in one launch == the same crops frame by frame (same kernel, bit-equal)"""
    from yolov7_tracker_amd import synth
    frames = torch.from_numpy(synth.make_frames(3, 40, 640, seq_idx=4)).cuda()
    rng = np.random.default_rng(6)
    xy, wh = rng.uniform(0, 500, (90, 2)), rng.uniform(10, 130, (90, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    idx = rng.integers(0, 3, 90).astype(np.int32)
    got = fused_extractor.features_for_frames(frames, boxes, idx).cpu()
    for f in range(3):
        sel = np.nonzero(idx == f)[0]
        assert torch.equal(got[sel], fused_extractor.features_for_boxes(frames[f], boxes[sel]).cpu())
    with pytest.raises(Exception):
        fused_extractor.features_for_frames(frames, boxes, idx[:5])


def test_fused_kernel_only_for_its_configuration():
    from yolov7_tracker_amd import _lib
    from yolov7_tracker_amd.tracker.reid import ReIDExtractor
    assert not ReIDExtractor(None, width=0.5, max_crops=4).fused            # other widths: the op list
    with pytest.raises(_lib.Y7TError):
        ReIDExtractor(None, size=(128, 256), max_crops=4, fused=True)


def test_deepsort_with_device_reid_tracks_the_scene(fused_extractor):
    """BASELINE config 4 in miniature: frames of the synthetic scene, its detections, appearance features from the device OSNet over the
    device crops.  With real (non-degenerate) embeddings the cascade keeps identities: almost every object holds one id over the clip."""
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.deepsort import DeepSORT
    n_frames, n_obj, size = 25, 30, 640
    frames = synth.make_frames(n_frames, n_obj, size, seq_idx=5)
    gt = []
    dets = synth.make_detections(n_frames, n_obj, size, seq_idx=1005, miss=0.0, fp=0.0, ground_truth=gt)     # same objects as make_frames(seq 5)
    BaseTrack._count = 0
    t = DeepSORT(types.SimpleNamespace(conf_thresh=0.05, track_buffer=30, kalman_format="default", img_size=size, iou_thresh=0.5), reid_model=fused_extractor)
    ids_per_frame = []
    for f in range(n_frames):
        cur = t.update(dets[f], torch.from_numpy(frames[f]))
        ids_per_frame.append(sorted(c.track_id for c in cur))
    assert len(ids_per_frame[-1]) >= 0.8 * len(dets[-1])
    assert BaseTrack._count <= 1.3 * n_obj                     # few identity switches / re-births
    assert len(set(ids_per_frame[5]) & set(ids_per_frame[-1])) >= 0.7 * len(ids_per_frame[5])


def test_extractor_from_torchreid_checkpoint(tmp_path, extractor):
    """a checkpoint in the form the reference loads (reid_models/load_model_tools.py: {'state_dict': ...} with 'module.' prefixes and a
    classifier head) gives the same network; DeepSORT picks it up from opts.reid_model_path like deepsort.py:14"""
    from yolov7_tracker_amd.tracker.reid import ReIDExtractor
    from yolov7_tracker_amd.tracker.deepsort import DeepSORT
    sd = {"module." + k: v for k, v in extractor.sd.items()}
    sd["module.classifier.weight"], sd["module.classifier.bias"] = torch.zeros(1, 512), torch.zeros(1)
    path = str(tmp_path / "osnet_x0_25.pth")
    torch.save({"state_dict": sd, "epoch": 3}, path)
    e2 = ReIDExtractor.from_checkpoint(path, max_crops=8)
    x = torch.randn((4, 128, 64, 3), generator=torch.Generator().manual_seed(2))
    assert torch.equal(e2.forward_crops(x), extractor.forward_crops(x))
    t = DeepSORT(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=640, iou_thresh=0.5, reid_model_path=path))
    assert isinstance(t.reid_model, ReIDExtractor)


def test_deepsort_embedding_network_matches_oracle(tmp_path):
    """the reference's OWN DeepSORT embedding network (reid_models/deepsort_reid.py Net(reid=True), what its Extractor loads from weights/ckpt.t7)
    against oracle/reid_torch.py::deepsort_net_forward (== the reference's class bit for bit, tests/test_reid_oracle.py).  Two device paths: the
    fp32 op list (BatchNorm folded: 2e-4 of the feature scale) and the MFMA path -- fp16 storage / fp32 accumulate on the detector's conv kernels:
    17 stored tensors on the longest path, each rounded at 2^-11 -> 3e-3 of the feature scale (the CPU emulation of the same roundings gives 4e-4),
    cosine >= 1 - 1e-5.  40 crops so that the 64-channel stage takes the LDS-patch kernel and the 512-channel stage split-K.  Then from a
    checkpoint in the reference's format through DeepSORT's reid_model_path like deepsort.py:14"""
    from oracle import reid_torch
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.tracker import reid
    from yolov7_tracker_amd.tracker.deepsort import DeepSORT
    sd = reid.deepsort_net_random_state_dict(7)
    x = torch.randn((40, 3, 128, 64), generator=torch.Generator().manual_seed(8))
    want = reid_torch.deepsort_net_forward(sd, x)
    scale = float(want.abs().max())
    exact = reid.ReIDExtractor(sd, max_crops=8, mfma=False)
    assert exact.arch == "deepsort" and not exact.fused and not exact.mfma and exact.feat_dim == 512
    got = exact.forward_crops(x[:5].permute(0, 2, 3, 1).contiguous()).cpu()
    np.testing.assert_allclose(got.norm(dim=1).numpy(), 1.0, rtol=1e-5)
    assert float((got - want[:5]).abs().max()) <= 2e-4 * scale, float((got - want[:5]).abs().max()) / scale
    e = reid.ReIDExtractor(sd, max_crops=48)
    assert e.arch == "deepsort" and e.mfma
    got = e.forward_crops(x.permute(0, 2, 3, 1).contiguous()).cpu()
    err = float((got - want).abs().max()) / scale
    cos = ((got * want).sum(1) / (got.norm(dim=1) * want.norm(dim=1))).min().item()
    print("deepsort Net on the MFMA kernels: max err %.2e of the feature scale, min cosine 1 - %.1e" % (err, 1 - cos))
    assert err <= 3e-3 and cos >= 1 - 1e-5, (err, cos)
    np.testing.assert_allclose(got.norm(dim=1).numpy(), 1.0, rtol=1e-5)
    assert torch.equal(e.forward_crops(x[:3].permute(0, 2, 3, 1).contiguous()).cpu(), got[:3]) or \
        float((e.forward_crops(x[:3].permute(0, 2, 3, 1).contiguous()).cpu() - got[:3]).abs().max()) <= 3e-3 * scale     # (another batch size may pick other kernels)
    # crops from a frame: the same preprocessing as the OSNet path (Extractor._preprocess)
    frame = synth.make_frames(1, 80, 640, seq_idx=3)[0]
    boxes = np.array([[10, 20, 60, 150], [100.7, 0.2, 130.9, 64.5], [300, 300, 364, 428]], np.float32)
    want_b = reid_torch.deepsort_net_forward(sd, reid_torch.preprocess(frame, boxes))
    got_b = e.features_for_boxes(frame, boxes).cpu()
    assert float((got_b - want_b).abs().max()) <= 3e-3 * float(want_b.abs().max())
    assert float((exact.features_for_boxes(frame, boxes).cpu() - want_b).abs().max()) <= 1e-3 * float(want_b.abs().max())
    # checkpoint in the reference's format ({'net_dict': ...} with the classifier head the reid=True forward never runs)
    ck = dict(sd)
    ck["classifier.0.weight"], ck["classifier.0.bias"] = torch.zeros(256, 512), torch.zeros(256)
    path = str(tmp_path / "ckpt.t7")
    torch.save({"net_dict": ck, "acc": 0.9, "epoch": 40}, path)
    t = DeepSORT(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=640, iou_thresh=0.5, reid_model_path=path))
    assert isinstance(t.reid_model, reid.ReIDExtractor) and t.reid_model.arch == "deepsort" and t.reid_model.mfma
    assert torch.equal(t.reid_model.features_for_boxes(frame, boxes).cpu(), got_b)
