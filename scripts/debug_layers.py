"""per-layer error of the HIP detector vs the fp16-emulating oracle (debug aid)"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import detector_torch as dt
from yolov7_tracker_amd.detector import arch, model
name = sys.argv[1] if len(sys.argv) > 1 else "yolov7-tiny"
nc, hw, B = (80, (128, 192), 2) if "tiny" in name else (10, (256, 320), 2)
det = model.Detector(arch.ARCHS[name](nc), None, img_size=hw, max_batch=B)
img = torch.rand((B, 3) + hw, generator=torch.Generator().manual_seed(1))
out = det(img)[0]
torch.cuda.synchronize()
_, _, vals = dt.forward(det.nodes, det._sd, img, det.spec["anchors"], keep=True, fp16=True)
p = det.plan
for n in p.nodes:
    if n.kind in ("input", "detect") or n.home is None or n.idx not in vals:
        continue
    ref = vals[n.idx].permute(0, 2, 3, 1)   # B,H,W,C
    ld = n.ld
    got = det.buffer_view(n.home, B, ld).view(B, n.h, n.w, ld)[..., n.coff:n.coff + n.c].float().cpu()
    if n.kind == "reorg":
        got = got[..., :12]
    err = (got - ref).abs()
    print("%3d %-7s layer %3d  %4dx%-4d c=%4d  std %.3f  mean err %.2e  max err %.2e  rel(mean/std) %.2e" % (
        n.idx, n.kind, n.layer, n.h, n.w, n.c, ref.std().item(), err.mean().item(), err.max().item(), err.mean().item() / (ref.std().item() + 1e-9)))
