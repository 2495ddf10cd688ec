"""Per-sequence frame source with the reference's contract (/root/reference/tracker/tracker_dataloader.py:20-134):
`__getitem__ -> (img (3,h',w') float32 RGB /255 letterboxed, ori_img (H,W,3) uint8 BGR)`.

cv2 is not part of this environment, so files are decoded with PIL and the letterbox resize is a numpy two-tap bilinear filter
with cv2.INTER_LINEAR's geometry (half-pixel centres, edge-clamped taps, NO anti-aliasing when shrinking -- PIL's BILINEAR
widens its support there and is a different filter); the same arithmetic as the device kernel k_letterbox_layout, so
--device_preprocess and the host loader feed the network the same pixels.  Native-size frames (the synthetic 1280x1280
sequences of BASELINE configs) need neither resize nor padding and go through unchanged.
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
        img = _resize_linear(img, new_unpad[1], new_unpad[0])
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), np.uint8)
    out[...] = np.asarray(color, np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out, (r, r), (dw, dh)


def _resize_linear(img, new_h, new_w):
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) geometry in float32: source coordinate (d + 0.5) * scale - 0.5,
    taps clamped to the image, result rounded to uint8 (OpenCV's 8-bit path uses 11-bit fixed-point weights: +-1 grey level on ties)"""
    h0, w0 = img.shape[:2]
    one, half = np.float32(1), np.float32(0.5)
    sy = (np.arange(new_h, dtype=np.float32) + half) * np.float32(np.float32(h0) / np.float32(new_h)) - half
    sx = (np.arange(new_w, dtype=np.float32) + half) * np.float32(np.float32(w0) / np.float32(new_w)) - half
    iy, ix = np.floor(sy).astype(np.int64), np.floor(sx).astype(np.int64)
    ay = (sy - iy.astype(np.float32)).reshape(-1, 1, 1)
    ax = (sx - ix.astype(np.float32)).reshape(1, -1, 1)
    iy1, ix1 = np.clip(iy + 1, 0, h0 - 1), np.clip(ix + 1, 0, w0 - 1)
    iy, ix = np.clip(iy, 0, h0 - 1), np.clip(ix, 0, w0 - 1)
    src = img.astype(np.float32)
    top = (one - ax) * src[iy][:, ix] + ax * src[iy][:, ix1]
    bot = (one - ax) * src[iy1][:, ix] + ax * src[iy1][:, ix1]
    return np.rint((one - ay) * top + ay * bot).astype(np.uint8)


class TrackerLoader(torch.utils.data.Dataset):
    def __init__(self, path, img_size=1280, format='origin', seq=None, pre_process_method='v7', model_stride=32, device_preprocess=False,
                 yolo_data_root=''):
        """tracker_dataloader.py:21-62.  format 'origin': `path` is one sequence's folder of frames.  format 'yolo': `path` is the path FILE (test.txt) whose lines
        are image paths relative to the data root; the frames of sequence `seq` are the lines whose parent folder name is contained in `seq` (the reference's
        `elems[-2] in seq`, :50).  The reference hard-codes its data root ('/data/wujiapeng/datasets/', :33); here it comes from the dataset yaml
        (YOLO_DATA_ROOT, default DATASET_ROOT)."""
        super().__init__()
        self.device_preprocess = device_preprocess   # True: hand over only the raw frame; letterbox runs on the GPU (Detector.forward_frames)
        self.DATA_ROOT = yolo_data_root if format == 'yolo' else path
        self.format, self.pre_process_method, self.model_stride = format, pre_process_method, model_stride
        self.img_files = []
        if format == 'origin':
            assert os.path.isdir(path), f'your path is {path}, path must be your dataset path'
            self.img_files = sorted(os.listdir(path))
        elif format == 'yolo':
            assert os.path.isfile(path), f'your path is {path}, path must be your path file'
            with open(path, 'r') as f:
                for line in f.readlines():
                    line = line.strip()
                    elems = line.split('/')
                    if len(elems) >= 2 and elems[-2] in seq:
                        self.img_files.append(os.path.join(self.DATA_ROOT, line))      # absolute path (os.path.join keeps an absolute `line` as it is)
        else:
            raise NotImplementedError("data_format %r" % format)
        if isinstance(img_size, int):
            self.width, self.height = img_size, img_size
        else:
            self.width, self.height = img_size[0], img_size[1]

    def __getitem__(self, index):
        from PIL import Image
        p = os.path.join(self.DATA_ROOT, self.img_files[index]) if self.format == 'origin' else self.img_files[index]
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
