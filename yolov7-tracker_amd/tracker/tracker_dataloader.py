"""Per-sequence frame source with the reference's contract (/root/reference/tracker/tracker_dataloader.py:20-134):
`__getitem__ -> (img (3,h',w') float32 RGB /255 letterboxed, ori_img (H,W,3) uint8 BGR)`.

cv2 is not part of this environment, so files are decoded with PIL and the letterbox resize is PIL's bilinear filter
(the reference: cv2.INTER_LINEAR -- same kernel, not bit-identical at the edges).  Native-size frames (the synthetic
1280x1280 sequences of BASELINE configs) need neither resize nor padding and go through unchanged.
`SyntheticLoader` serves the seeded synthetic sequences (yolov7_tracker_amd.synth) without touching the disk."""
import os

import numpy as np
import torch


def letterbox(img, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleup=True, stride=32):
    """tracker_dataloader.py:100-130"""
    shape = img.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    dw /= 2
    dh /= 2
    if shape[::-1] != new_unpad:
        from PIL import Image
        img = np.asarray(Image.fromarray(img).resize(new_unpad, Image.BILINEAR))
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), np.uint8)
    out[...] = np.asarray(color, np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out, (r, r), (dw, dh)


class TrackerLoader(torch.utils.data.Dataset):
    def __init__(self, path, img_size=1280, format='origin', seq=None, pre_process_method='v7', model_stride=32, device_preprocess=False):
        super().__init__()
        self.device_preprocess = device_preprocess   # True: hand over only the raw frame; letterbox runs on the GPU (Detector.forward_frames)
        self.DATA_ROOT = path
        self.format, self.pre_process_method, self.model_stride = format, pre_process_method, model_stride
        if format != 'origin':
            raise NotImplementedError("data_format %r (only 'origin' folders of frames)" % format)
        assert os.path.isdir(path), f'your path is {path}, path must be your dataset path'
        self.img_files = sorted(os.listdir(path))
        if isinstance(img_size, int):
            self.width, self.height = img_size, img_size
        else:
            self.width, self.height = img_size[0], img_size[1]

    def __getitem__(self, index):
        from PIL import Image
        p = os.path.join(self.DATA_ROOT, self.img_files[index])
        ori_img = np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1].copy()   # (H, W, C) BGR like cv2.imread
        if self.device_preprocess:
            return torch.empty(0), torch.from_numpy(ori_img)
        img = letterbox(ori_img, new_shape=(self.height, self.width), stride=self.model_stride)[0]
        img = np.ascontiguousarray(img[:, :, ::-1].transpose(2, 0, 1))         # BGR to RGB, HWC to CHW
        img = torch.from_numpy(img).float()
        img /= 255.0
        return img, torch.from_numpy(ori_img)

    def __len__(self):
        return len(self.img_files)


class SyntheticLoader(torch.utils.data.Dataset):
    """seeded synthetic sequence: frames rendered by synth.make_frames, plus the scene's detections (used as the
    tracker's input with --synthetic_dets, since random detector weights do not detect anything meaningful)."""

    def __init__(self, n_frames, n_obj, size, seq_idx, device_preprocess=False):
        from .. import synth
        self.device_preprocess = device_preprocess
        self.frames = synth.make_frames(n_frames, n_obj, size, seq_idx)
        self.dets = synth.make_detections(n_frames, n_obj, size, seq_idx)

    def __getitem__(self, i):
        ori = self.frames[i]
        if self.device_preprocess:
            return torch.empty(0), torch.from_numpy(ori)
        img = torch.from_numpy(np.ascontiguousarray(ori[:, :, ::-1].transpose(2, 0, 1))).float()
        img /= 255.0
        return img, torch.from_numpy(ori)

    def __len__(self):
        return len(self.frames)
