"""Detector half of the hot path: YOLOv7 graph lowering + weight packing (host, Python) over the HIP kernels of
liby7t.so (implicit-GEMM MFMA conv, pooling, decode + NMS)."""
from .model import Detector, attempt_load, load_checkpoint, checkpoint_anchors, non_max_suppression, scale_coords, check_img_size  # noqa: F401
