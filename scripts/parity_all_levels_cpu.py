"""CPU exploration (no GPU): SURVEY 8a's candidate bar with ALL FOUR Detect levels live and UNDAMPED width / height logits, fp32 oracle against the oracle's own
fp16-storage emulation (oracle/detector_torch.py forward(fp16=True): fp16 weights / activations, fp32 accumulate -- the arithmetic of the HIP path), one 1280 x 1280
frame.  Says what the device test of this configuration (tests/test_detector_pinned_gpu.py::test_all_levels_*) should expect before GPU minutes are spent on it.

    python scripts/parity_all_levels_cpu.py [damp] [level_offsets as a,b,c,d]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import detector_torch as dt          # noqa: E402
from yolov7_tracker_amd import synth             # noqa: E402
from yolov7_tracker_amd.detector import arch, graph, weights  # noqa: E402


def main():
    damp = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    lo = [float(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.0] * 4
    torch.set_num_threads(os.cpu_count())
    nc, H = 10, 1280
    spec = arch.ARCHS["yolov7-w6"](nc)
    frames = synth.make_frames(1, 80, H, seq_idx=0)
    img = (torch.from_numpy(frames[:1][..., ::-1].copy()).permute(0, 3, 1, 2).float() / 255.0).contiguous()
    nodes, _ = graph.parse(spec)
    plan = graph.lower(graph.parse(spec)[0], H, H, 1)
    sd = weights.calibrate_bn(nodes, weights.random_state_dict(plan.wlayout, 0, bn_bias_mean=2.0), seed=0, image=img)
    na, no = 3, nc + 5
    if damp != 1.0:
        for k in list(sd):
            if ".m." in k and k.endswith(".weight"):
                w = sd[k].clone().view(na, no, -1)
                w[:, 2:4] *= damp
                sd[k] = w.view(na * no, -1, 1, 1)
    # plant_objectness_bias (detector/model.py) restated on the fp16-emulating oracle's heads
    _, raw16 = dt.forward(nodes, sd, img, spec["anchors"], fp16=True)
    logits = torch.cat([r[0, ..., 4].reshape(-1) + lo[l] for l, r in enumerate(raw16)])
    q = torch.quantile(logits.float(), 1.0 - 2000 / logits.numel()).item()
    shift = float(np.log(0.01 / 0.99)) - q
    base = "model.%d" % next(n for n in nodes if n.kind == "detect").layer
    for l in range(4):
        b = sd["%s.m.%d.bias" % (base, l)].float().clone().view(na, no)
        b[:, 4] += shift + lo[l]
        b[:, 5:] += 4.0
        sd["%s.m.%d.bias" % (base, l)] = b.view(-1)
    dec32, raw32 = dt.forward(nodes, sd, img, spec["anchors"])
    dec16, raw16 = dt.forward(nodes, sd, img, spec["anchors"], fp16=True)
    for l, (a, b) in enumerate(zip(raw16, raw32)):
        e = (a - b).abs()
        print("level %d: logit std %.2f  mean/max |err| %.3e %.3e   wh-logit std %.2f  mean |err| on wh rows %.3e" % (
            l, b.std().item(), e.mean().item(), e.max().item(), b[..., 2:4].std().item(), e[..., 2:4].mean().item()))
    got, want = dt.candidates(dec16[0], 0.01), dt.candidates(dec32[0], 0.01)
    st = dt.compare_candidate_sets(got, want, 0.01, px=1.0, dconf=5e-3)
    print(st)
    rows0 = np.cumsum([0] + [int(r.shape[1] * r.shape[2] * r.shape[3]) for r in raw32])
    both = sorted(set(got) & set(want))
    lvl = np.searchsorted(rows0, np.array(both), side="right") - 1
    for l in range(4):
        rs = [r for r, v in zip(both, lvl) if v == l]
        if not rs:
            print("level %d: no candidates" % l)
            continue
        dc = np.array([np.abs(got[r][0] - want[r][0]).max() for r in rs])
        iou = np.array([dt.box_iou_1(got[r][0], want[r][0]) for r in rs])
        side = np.array([max(want[r][0][2] - want[r][0][0], want[r][0][3] - want[r][0][1]) for r in rs])
        ds = np.array([abs(got[r][1] - want[r][1]) for r in rs])
        ok = ((dc <= 1.0) | (iou >= 0.99)) & (ds <= 5e-3)
        print("level %d: %5d candidates, sides %.0f..%.0f px (median %.0f)  max dcoord %.2f px  min IoU %.4f  max dconf %.2e  within (1 px OR IoU >= 0.99) & 5e-3: %d / %d" % (
            l, len(rs), side.min(), side.max(), np.median(side), dc.max(), iou.min(), ds.max(), int(ok.sum()), len(rs)))


if __name__ == "__main__":
    main()
