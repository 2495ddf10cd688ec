"""oracle/letterbox_np.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's frame pre-processing, /root/reference/tracker/tracker_dataloader.py:64-130:
`_letterbox` (scale ratio, `auto=True` padding to the stride multiple, cv2.resize INTER_LINEAR, cv2.copyMakeBorder 114) and the
BGR->RGB / CHW / float32 / 255 tail of `__getitem__`.

PARITY UNPINNED for the resize itself: cv2 is not installed here and the reference has no fixture for it.  This restates
cv2.INTER_LINEAR's geometry (half-pixel centres, edge-clamped taps) in float arithmetic with rounding to uint8; OpenCV's
own 8-bit path uses 11-bit fixed-point weights, which can differ from this by one grey level on rounding ties.
"""
import numpy as np


def letterbox(img, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleup=True, stride=32):
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
        img = resize_bilinear(img, new_unpad[1], new_unpad[0])
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), np.uint8)
    out[...] = np.asarray(color, np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out


def resize_bilinear(img, new_h, new_w):
    H0, W0 = img.shape[:2]
    f32 = np.float32
    fy = (np.arange(new_h, dtype=f32) + f32(0.5)) * f32(f32(H0) / f32(new_h)) - f32(0.5)
    fx = (np.arange(new_w, dtype=f32) + f32(0.5)) * f32(f32(W0) / f32(new_w)) - f32(0.5)
    y0, x0 = np.floor(fy).astype(np.int64), np.floor(fx).astype(np.int64)
    wy, wx = (fy - y0.astype(f32))[:, None, None], (fx - x0.astype(f32))[None, :, None]
    y1, x1 = np.clip(y0 + 1, 0, H0 - 1), np.clip(x0 + 1, 0, W0 - 1)
    y0, x0 = np.clip(y0, 0, H0 - 1), np.clip(x0, 0, W0 - 1)
    im = img.astype(f32)
    t0 = (f32(1) - wx) * im[y0][:, x0] + wx * im[y0][:, x1]
    t1 = (f32(1) - wx) * im[y1][:, x0] + wx * im[y1][:, x1]
    return np.rint((f32(1) - wy) * t0 + wy * t1).astype(np.uint8)


def to_model_input(img_bgr_u8):
    """__getitem__ tail: BGR HWC uint8 -> RGB CHW float32 / 255"""
    x = np.ascontiguousarray(img_bgr_u8[:, :, ::-1].transpose(2, 0, 1)).astype(np.float32)
    return x / np.float32(255.0)
