"""The reference's OWN code timed beside the CPU port (the oracle), on this host's cores -- SURVEY.md 8d / BASELINE.md section 3, VERDICT r2 item 6.

Build container only (needs /root/reference; the GPU box does not have it, which is why bench.py's `cpu_baseline` there is kind = "port").
    python scripts/cpu_reference_vs_port.py [frames] > profiles/r03_cpu_reference_vs_port.txt

  detector : /root/reference/models/yolo.py::Model('cfg/deploy/yolov7-w6.yaml', nc=10).fuse().eval() through oracle/ref_harness.py (stub modules for the
             packages the reference imports and this image lacks), FP32, batch 1, 1280x1280, with torch.no_grad() and without it (the reference CLI
             omits it, tracker/track.py:144)            vs   oracle/detector_torch.py::forward (the port bench.py times)
  NMS      : /root/reference/utils/general.py::non_max_suppression with torchvision.ops.nms supplied by the oracle's greedy restatement (torchvision is
             absent)                                    vs   oracle/detector_torch.py::non_max_suppression
  tracker  : /root/reference/tracker/bytetrack.py::ByteTrack.update (lap.lapjv / cython_bbox.bbox_overlaps supplied by the oracle's C restatements)
                                                        vs   oracle/tracker_np.py
Same seeded weights, same frame, same synthetic scene on both sides; outputs compared (they are equal: the port is pinned to the reference by the test-suite).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import detector_torch as dt, ref_harness, tracker_np  # noqa: E402
from yolov7_tracker_amd import synth  # noqa: E402
from yolov7_tracker_amd.detector import arch, graph, weights  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cores = os.cpu_count()
torch.set_num_threads(cores)
print("host: %d cores (torch threads %d), torch %s, numpy %s" % (cores, torch.get_num_threads(), torch.__version__, np.__version__))
if not ref_harness.available():
    raise SystemExit("/root/reference is not present on this machine")

nc, H = 10, 1280
spec = arch.yolov7_w6(nc)
nodes, _ = graph.parse(spec)
plan = graph.lower(graph.parse(spec)[0], H, H, 1)
frames = synth.make_frames(1, 80, H, seq_idx=0)
img = (torch.from_numpy(frames[:1][..., ::-1].copy()).permute(0, 3, 1, 2).float() / 255.0).contiguous()
sd = weights.calibrate_bn(nodes, weights.random_state_dict(plan.wlayout, 0), seed=0, image=img)

m = ref_harness.build_reference_model("cfg/deploy/yolov7-w6.yaml", nc)
missing, unexpected = m.load_state_dict(sd, strict=False)
assert not unexpected
m.fuse().eval()


def timeit(fn, n):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    return (time.perf_counter() - t0) / n, r


def ref_nograd():
    with torch.no_grad():
        return m(img)[0]


t_ref_ng, out_ref = timeit(ref_nograd, N)
t_ref_g, _ = timeit(lambda: m(img)[0], N)
t_port, (dec, _) = timeit(lambda: dt.forward(nodes, sd, img, spec["anchors"]), N)
# (the reference is timed after fuse(): BatchNorm folded into the convs, utils/torch_utils.py:181-201; the port applies BatchNorm unfused in fp32 -- the
#  values agree to rounding, not bit for bit; tests/test_detector_oracle.py pins the UNFUSED reference Model == the port bit for bit)
err = float((out_ref - dec).abs().max() / dec.abs().max())
print("detector, w6 @ 1280x1280, batch 1, fp32:   reference Model (fused, no_grad) %.3f s/frame | reference as the CLI runs it (no no_grad) %.3f s/frame | "
      "port %.3f s/frame      max |ref - port| / max|port| = %.1e" % (t_ref_ng, t_ref_g, t_port, err))

# NMS on ~2000 candidates (the bench's load)
ns = ref_harness.load_detector()
pred = dec.clone()
pred[..., 4] = 0.0
idx = torch.randperm(pred.shape[1], generator=torch.Generator().manual_seed(0))[:2000]
pred[0, idx, 4] = torch.rand(2000, generator=torch.Generator().manual_seed(1)) * 0.9 + 0.05
pred[..., 5:] = torch.rand(pred[..., 5:].shape, generator=torch.Generator().manual_seed(2))
tv_ops = sys.modules.get("torchvision.ops") or getattr(ns.general, "torchvision", None)
from oracle import cnative  # noqa: E402
nms_fn = lambda b, s, t: torch.from_numpy(cnative.nms(b.numpy(), s.numpy(), t))
ns.general.torchvision.ops.nms = nms_fn          # the one call the reference makes into torchvision on this path (general.py:679)
t_ref_nms, r_nms = timeit(lambda: ns.general.non_max_suppression(pred, 0.01, 0.45)[0], 10)
t_port_nms, p_nms = timeit(lambda: dt.non_max_suppression(pred, 0.01, 0.45)[0], 10)
print("non_max_suppression, 2000 candidates:       reference %.2f ms | port %.2f ms      outputs equal: %s" % (t_ref_nms * 1e3, t_port_nms * 1e3, bool(torch.equal(r_nms, p_nms))))

# tracker: 300 frames, ~80 objects (configs[1]) and 60 frames x 500 objects (configs[2] load)
for n_obj, n_frames in ((80, 300), (500, 60)):
    dets = synth.make_detections(n_frames, n_obj, H, seq_idx=0, bounce=True)
    t0 = time.perf_counter()
    ref_rows = ref_harness.run_reference_tracker("bytetrack", dets)
    t_ref_trk = (time.perf_counter() - t0) / n_frames
    t0 = time.perf_counter()
    port_rows = tracker_np.run("bytetrack", dets)
    t_port_trk = (time.perf_counter() - t0) / n_frames
    same = all([r[0] for r in a] == [r[0] for r in b] for a, b in zip(ref_rows, port_rows))
    print("ByteTrack.update, %3d objects, %3d frames:   reference %.2f ms/frame | port %.2f ms/frame (1 thread each)      ids equal: %s" % (
        n_obj, n_frames, t_ref_trk * 1e3, t_port_trk * 1e3, same))
    if n_obj == 80:
        e2e_ref = 1.0 / (t_ref_g + t_ref_nms + t_ref_trk)
        e2e_ref_ng = 1.0 / (t_ref_ng + t_ref_nms + t_ref_trk)
        e2e_port = 1.0 / (t_port + t_port_nms + t_port_trk)
print("end to end (detector + NMS + ByteTrack, 80 objects):   reference as its CLI runs it %.2f fps | reference with no_grad %.2f fps | port %.2f fps   on %d cores"
      % (e2e_ref, e2e_ref_ng, e2e_port, cores))
