"""CPU exploration (no GPU): what the chained detect -> NMS -> ByteTrack parity run (tests/test_detector_pinned_gpu.py::test_chained_*) should expect.  The fp32 oracle chain
against the oracle's own fp16-storage emulation of the device (oracle/detector_torch.py forward(fp16=True)) on the first N frames of the benchmarked scene, conditioned
weights, all four Detect levels live: per frame how the two (n, 6) hand-overs differ, then HOTA / IDF1 of the fp16 chain's tracks graded against the fp32 chain's.

    python scripts/chained_cpu.py [frames=12]
"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import chained, detector_torch as dt   # noqa: E402
from yolov7_tracker_amd import synth               # noqa: E402
from yolov7_tracker_amd.detector import arch, graph, weights  # noqa: E402

LEVEL_QUOTA = tuple(float(v) for v in os.environ.get("LEVELS", "0.55,0.2,0.15,0.1").split(","))
OBJ_GAIN = float(os.environ.get("OBJ_GAIN", "2.75"))      # objectness logits widened: ~2000 anchors above 0.01 AND ~100 above 0.3 (a head that is confident about some objects)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    torch.set_num_threads(os.cpu_count())
    nc, H = 10, 1280
    spec = arch.ARCHS["yolov7-w6"](nc)
    frames = synth.make_frames(n, 80, H, seq_idx=0)
    img = chained.images(frames[:1])
    nodes, _ = graph.parse(spec)
    plan = graph.lower(graph.parse(spec)[0], H, H, 1)
    sd = weights.calibrate_bn(nodes, weights.random_state_dict(plan.wlayout, 0, bn_bias_mean=2.0), seed=0, image=img)
    na, no = 3, nc + 5
    for k in list(sd):
        if ".m." in k and k.endswith(".weight"):
            w = sd[k].clone().view(na, no, -1)
            w[:, 2:4] *= 0.25
            w[:, 4] *= OBJ_GAIN
            sd[k] = w.view(na * no, -1, 1, 1)
    _, raw16 = dt.forward(nodes, sd, img, spec["anchors"], fp16=True)
    base = "model.%d" % next(nd for nd in nodes if nd.kind == "detect").layer
    for l, r in enumerate(raw16):      # Detector.plant_objectness_bias(level_quota=...) restated
        x = r[0, ..., 4].reshape(-1).float()
        shift = float(np.log(0.01 / 0.99)) - torch.quantile(x, max(0.0, 1.0 - LEVEL_QUOTA[l] * 2000 / x.numel())).item()
        b = sd["%s.m.%d.bias" % (base, l)].float().clone().view(na, no)
        b[:, 4] += shift
        b[:, 5:] += 4.0
        sd["%s.m.%d.bias" % (base, l)] = b.view(-1)
    d32, d16 = [], []
    for f in range(n):
        im = chained.images(frames[f:f + 1])
        for fp16, out in ((False, d32), (True, d16)):
            dec, _ = dt.forward(nodes, sd, im, spec["anchors"], fp16=fp16)
            r = dt.non_max_suppression(dec, 0.01, 0.45)[0].clone()
            r[:, :4] = dt.scale_coords_round((H, H), r[:, :4], (H, H))
            out.append(r.numpy().astype(np.float32))
        a, b = d32[-1], d16[-1]
        oa, ob = chained.detection_set_difference(a, b)
        print("frame %2d: fp32 %d rows, fp16 %d rows, identical hand-over %s; only fp32 %d (conf %s), only fp16 %d (conf %s); rows with conf >= 0.2: %d / %d, exact same hand-over: %s"
              % (f, len(a), len(b), chained.same_detections(a, b), len(oa), np.round(a[oa, 4], 3).tolist()[:6], len(ob), np.round(b[ob, 4], 3).tolist()[:6],
                 int((a[:, 4] >= 0.2).sum()), int((b[:, 4] >= 0.2).sum()), chained.same_detections(a, b)), flush=True)
    t32, t16 = chained.track("bytetrack", d32), chained.track("bytetrack", d16)
    print("tracks per frame fp32:", [len(x) for x in t32])
    print("tracks per frame fp16:", [len(x) for x in t16])
    with tempfile.TemporaryDirectory() as tmp:
        print(chained.grade(tmp, t32, t16))


if __name__ == "__main__":
    main()
