"""the reference's DeepSORT embedding network (reid_models/deepsort_reid.py Net) on the device: timing of the fp16 MFMA op list (and the fp32 one),
and -- with a path argument -- the features of a fixed input saved for comparing runs (Y7T_REID_PATCH=0 / 1 are separate processes)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd.tracker import reid  # noqa: E402

tag = "patch kernel allowed" if os.environ.get("Y7T_REID_PATCH") == "1" else "generic kernel"
for n, mf in ((80, True), (2560, True)) + (((80, False),) if "--fp32" in sys.argv else ()):
    e = reid.ReIDExtractor(None, arch="deepsort", max_crops=n, mfma=mf)
    x = torch.randn((n, 128, 64, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    for _ in range(3):
        out = e.forward_crops(x)
    torch.cuda.synchronize()
    t0, k = time.time(), (10 if mf else 3)
    for _ in range(k):
        e.forward_crops(x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / k
    print("deepsort_reid.Net, %s: %d crops %.2f ms (%.1f TFLOP/s of 2.2 GFLOP per crop)" % (("fp16 MFMA op list, " + tag) if mf else "fp32 op list", n, dt * 1e3,
                                                                                          n * 2.2e9 / dt / 1e12), flush=True)
    if mf and n == 2560 and len(sys.argv) > 1 and not sys.argv[-1].startswith("--"):
        torch.save(out.cpu(), sys.argv[-1])
    del e
