"""GPU: the detector pinned at the BENCHMARKED configuration -- YOLOv7-w6 @ 1280x1280, nc = 10, 80 frames per forward (bench.py's
default), i.e. the launch list bench.py times: LDS-patch / multi-tile / strip kernels, 256-pixel tiles, panel-packed 1x1 layers.

  (a) per op, teacher-forced: after ONE forward of the whole list every tensor is still in the arena (one buffer per tensor), so for every
      op the test reads the op's actual fp16 input slice, evaluates the oracle's layer on it (oracle/detector_torch.py: BN fold in float64,
      fp16 weights, fp32 accumulate -- /root/reference/models/common.py:99-111, utils/torch_utils.py:181-201) and compares with the op's
      actual output slice at the LAYER tolerance (rtol 6e-4 / atol 3e-4: half an fp16 ulp + summation order); pools, upsamples and
      concat copies must be bit-exact.  No chaos argument: every op is judged on identical inputs.
      The tolerance is a forward error bound, not a fudge: |got - ref| <= half an fp16 ulp of the result (rtol 6e-4 / atol 3e-4 covers it)
      + 2 sqrt(K) u32 * sum_k |w_k x_k| for the fp32 accumulation of K terms in another order (u32 = 2^-24; the probabilistic forward
      error bound -- the worst-case one has K in place of 2 sqrt(K) -- matters only where the seeded network's activations are large).
  (b) raw Detect heads end to end (image -> 4 head tensors) against the fp32 oracle.
  (c) image -> boxes: decode + NMS + scale_coords + round of the fp16 network against the fp32 oracle's, at SURVEY.md 8a's box-level
      bar (same count, same class, |dcoord| <= 1 px after round, |dconf| <= 5e-3).
  (b) and (c) need a network that does not amplify rounding noise.  iid zero-mean random weights + BatchNorm sit on the chaotic side of
  the order/chaos transition (relative perturbation x ~1.1 per SiLU layer, x ~300 over w6's depth: measured 3-7 % of the logit spread
  at this size, test (b0) below keeps that as a loose sanity bound); BatchNorm shifts of ~ +2 (weights.random_state_dict bn_bias_mean)
  put the SiLUs in their near-linear region, the growth factor drops to ~1.0 and the random network is as well-conditioned as a
  trained one.  Same kernels, same launch list -- only the numbers in the weight blob differ.
"""
import collections
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

B_BENCH = 80                    # == bench.DEFAULT_BATCH (asserted below): the frames per forward of the timed launch list
# the score noise the kept-set explanations may invoke is an A-PRIORI bound, not the run's own maximum (a larger kernel error must not widen its own
# tolerance): SURVEY 8a's confidence bar itself, 5e-3 (~60 fp16-stored tensors in a row predict 0.2 % of the logit spread = ~2e-3 in confidence at the sigmoid's steepest
# point; the worst values measured on this network: 3.3e-3 over two frames, 4.3e-3 over the 32 frames of tests/test_chained_gpu.py)
SCORE_NOISE = 5e-3
CHECK_FRAMES = [0, 13, B_BENCH // 2, B_BENCH - 1]      # frames whose every pixel is compared (first / middle / last M rows of every tile grid; B // 2: the first frame of
                                                      # the second run of frames of the layers whose tensors pass 2 GiB -- csrc/y7t_detector.hip::forward_impl)


@pytest.fixture(scope="module")
def bench_det():
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.detector import arch, model
    det = model.Detector(arch.ARCHS["yolov7-w6"](10), None, img_size=(1280, 1280), max_batch=B_BENCH, seed=0)
    frames_host = synth.make_frames(B_BENCH, 80, 1280, seq_idx=0)      # what bench.py feeds
    frames = torch.from_numpy(frames_host).cuda()
    import bench
    bench.plant_objectness_bias(det, frames)                             # as bench.py does: every Detect level supplies its quota of the candidates
    out = det(frames)[0]                                                 # y7t_input_layout + y7t_det_forward: the timed launch list
    torch.cuda.synchronize()
    return det, frames_host, out


def _conditioned_detector(damp_wh, obj_gain=1.0, level_quota=None):
    """the benchmarked launch list on well-conditioned weights (BatchNorm shifts ~ +2, statistics calibrated on frame 0 of the scene); ALL FOUR Detect levels
    live (no per-level objectness offsets: ~2000 candidates per frame wherever the head puts them).  damp_wh: factor on the width / height rows of the Detect
    convs -- 0.25 gives boxes of roughly anchor size (19 ... 1000 px: what a trained head produces), 1.0 leaves the random head's logits as they are
    (sigmoids saturate: boxes of up to 4 x the anchor, 2000+ px on the coarse levels).  obj_gain: factor on the objectness rows (a head that is CONFIDENT about some of
    its ~2000 candidates -- what the tracker's 0.2 / 0.3 thresholds need: tests/test_chained_gpu.py); level_quota: share of the candidates per Detect level (default bench.LEVEL_QUOTA)"""
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.detector import arch, graph, model, weights
    spec = arch.ARCHS["yolov7-w6"](10)
    frames_host = synth.make_frames(B_BENCH, 80, 1280, seq_idx=0)
    cal = torch.from_numpy(frames_host[:1][..., ::-1].copy()).permute(0, 3, 1, 2).float() / 255.0
    nodes, _ = graph.parse(spec)
    plan = graph.lower(graph.parse(spec)[0], 1280, 1280, 1)
    sd = weights.calibrate_bn(nodes, weights.random_state_dict(plan.wlayout, 0, bn_bias_mean=2.0), seed=0, image=cal)
    if damp_wh != 1.0 or obj_gain != 1.0:
        for k in list(sd):
            if ".m." in k and k.endswith(".weight"):                 # Detect 1x1 convs: rows (anchor, [x, y, w, h, obj, cls...])
                w = sd[k].clone().view(3, 15, -1)
                w[:, 2:4] *= damp_wh
                w[:, 4] *= obj_gain
                sd[k] = w.view(45, -1, 1, 1)
    det = model.Detector(spec, sd, img_size=(1280, 1280), max_batch=B_BENCH)
    frames = torch.from_numpy(frames_host).cuda()
    import bench
    if level_quota is None:
        bench.plant_objectness_bias(det, frames)                         # all four Detect levels live (bench.LEVEL_QUOTA)
    else:
        det.plant_objectness_bias(frames, 2000, level_quota=level_quota)
    out = det(frames)[0]
    torch.cuda.synchronize()
    return det, frames_host, out


@pytest.fixture(scope="module")
def smooth_det():
    """anchor-sized boxes (width / height logits x 0.25), all four Detect levels live: the weights bench.py times (`--weights conditioned`)"""
    return _conditioned_detector(0.25)


@pytest.fixture(scope="module")
def all_levels_det():
    """all four Detect levels live AND undamped width / height logits"""
    return _conditioned_detector(1.0)


def _slice(det, buf, ld, coff, c, H, W, frames):
    v = det.buffer_view(buf, B_BENCH, ld).view(B_BENCH, H, W, ld)
    return v[frames][..., coff:coff + c]


def test_launch_list_is_the_benchmarked_one(bench_det):
    import bench
    assert bench.DEFAULT_BATCH == B_BENCH
    det, _, _ = bench_det
    names = det.launch_list(B_BENCH)
    hist = collections.Counter(names)
    print("launch list @ B=%d:" % B_BENCH, dict(hist))
    fam = collections.Counter(n.split("<")[0] for n in names)
    # the kernel families 45 % of the bench's conv time runs on (profiles/r01_bench_kernel_stats.csv) are all in this list
    assert fam["patch"] + fam["ws128"] >= 10 and (fam["patch_mt"] >= 4 or fam["ws64"] == 7) and fam["patch_strip"] >= 8, fam
    if os.environ.get("Y7T_CONV_WS128", "1") != "0":                     # the 128 -> 128 k layers with the filter bank in registers, tiles from the op's tile counter (round 5)
        assert fam["ws128"] == 11 and all(n.endswith(" dyn") for n in names if n.startswith(("ws64", "ws128"))), hist
    # the stride-2 patch kernel where it measured faster than the generic kernel (four of the eight down-sampling layers; detector/graph.py::patch_s2_eligible)
    ws128_on = os.environ.get("Y7T_CONV_WS128", "1") != "0"
    assert fam["patch_s2"] == {"auto": 2 if ws128_on else 4, "0": 0, "1": 6 if ws128_on else 8}[os.environ.get("Y7T_CONV_PATCH_S2", "auto")], fam      # (the two Cin = 128 layers are on ws128's stride-2 form)
    assert fam["ws128_s2"] == (2 if ws128_on else 0), fam
    if os.environ.get("Y7T_CONV_WS_S2", "1") != "0":
        assert names[1] == ("ws_s2<2,32> + 1x1" if os.environ.get("Y7T_CONV_WS_S2_FUSE", "1") != "0" else "ws_s2<2,32>"), names[1]      # the 640^2 64 -> 128 stride-2 layer: filter bank in registers
    else:
        assert any(n.startswith("igemm<256,") for n in names) or os.environ.get("Y7T_CONV_PATCH_S2") == "1", hist     # 256-pixel tiles
    if os.environ.get("Y7T_CONV_WS", "1") != "0":                        # the 64 -> 64 layers with the filter bank in registers
        assert fam["ws64"] == 7 and fam["patch_mt"] == 0, fam
    if os.environ.get("Y7T_CONV_P8", "1") not in ("0", "all"):           # the 1x1 layers with Cout % 256 == 0 where the 256 x 256 x 64 ping-pong pipeline measured faster (Cin >= 1024, >= 1500 tiles, or Cin >= 512 on a grid that fills its rounds: detector/graph.py::p8_eligible)
        assert fam["p8"] == 23, fam
    assert names[0] == "stem_u8<direct>", names[0]                       # uint8 frame -> stem conv in one kernel
    assert any(n.startswith("igemm<128,128,32,2> 1x1") for n in names), hist
    assert sum("upsample-on-read" in n for n in names) == 3 and "upsample2x" not in names, hist
    det(torch.from_numpy(bench_det[1]).cuda())       # launch_list re-ran the ops in place; leave the arena as a clean forward
    torch.cuda.synchronize()


def test_every_op_matches_the_oracle_teacher_forced(bench_det):
    from oracle import detector_torch as dt
    det, frames_host, _ = bench_det
    p = det.plan
    sd = det._sd
    torch.set_num_threads(min(32, torch.get_num_threads() if torch.get_num_threads() > 1 else 32))
    fr = CHECK_FRAMES
    # op 0's input: the layout kernel (BGR -> RGB, /255, ReOrg, fp16) against the oracle's ReOrg of the float image
    img = torch.from_numpy(frames_host[fr][..., ::-1].copy()).permute(0, 3, 1, 2).float() / 255.0     # tracker_dataloader.py:83-88
    re = torch.cat([img[..., ::2, ::2], img[..., 1::2, ::2], img[..., ::2, 1::2], img[..., 1::2, 1::2]], 1)   # models/common.py:48-53
    re16 = re.half().float()                                        # the fp16 values the stem convolves
    if not p.stem_fused:                                            # (fused stem: the layout tensor is never written -- the stem's OUTPUT is checked below)
        got0 = _slice(det, 0, p.in_ld, 0, p.in_ld, 640, 640, fr).float().cpu()
        assert torch.equal(got0[..., :12], re16.permute(0, 2, 3, 1)) and float(got0[..., 12:].abs().max()) == 0.0
    ci = 0
    names = det.launch_list(B_BENCH)
    det(torch.from_numpy(frames_host).cuda())
    torch.cuda.synchronize()
    worst = collections.defaultdict(float)
    n_conv = n_other = n_up = 0
    for oi, op in enumerate(p.ops):
        H, W, Cin = int(op["H"]), int(op["W"]), int(op["Cin"])
        if oi == 0 and int(op["in_buf"]) == 0:
            x = re16.clone()                                        # op 0 reads the frame: BGR -> RGB, /255, ReOrg, fp16 (whether or not that tensor exists in HBM)
        else:
            x = _slice(det, int(op["in_buf"]), int(op["in_ld"]), int(op["in_coff"]), Cin, H, W, fr).float().cpu().permute(0, 3, 1, 2).contiguous()
        if int(op["up_C"]) > 0:      # upsample-on-read: these channels of the concat exist only at half resolution (nn.Upsample(None, 2, 'nearest'))
            c0, cu = int(op["up_c0"]), int(op["up_C"])
            lo = _slice(det, int(op["up_buf"]), int(op["up_ld"]), int(op["up_coff"]), cu, H // 2, W // 2, fr).float().cpu().permute(0, 3, 1, 2)
            x[:, c0:c0 + cu] = F.interpolate(lo, scale_factor=2, mode="nearest")
            n_up += 1
        if int(op["type"]) == 0:
            wl = p.wlayout[ci]
            ci += 1
            x = x[:, :wl["cin"]]                                   # the stem's 12 real channels of the 16-channel layout
            k, s_, pd = int(op["KH"]), int(op["stride"]), int(op["pad"])
            extra_tol = 0.0
            if wl["kind"] == "conv" and wl.get("fused_next"):      # korder 11: the stride-2 layer + the twin 1x1 behind it in one launch; the tensor between them is never written
                w2 = p.wlayout[ci]
                ci += 1
                assert int(op["korder"]) == 11 and w2.get("fused_prev") and isinstance(w2["wkey"], tuple)
                mid = dt._conv_bn_act(x, sd, wl["wkey"], k, s_, pd, wl["act"], fp16=True, round_out=True)        # (fp16 in LDS, as it would be in memory)
                ref = torch.cat([dt._conv_bn_act(mid, sd, key, 1, 1, 0, w2["act"], fp16=True, round_out=False) for key in w2["wkey"]], 1)
                absum = torch.cat([dt.conv_abs_sum(mid, sd, key, 1, 0) for key in w2["wkey"]], 1)
                extra_tol = 2.0 ** -11      # a 1-ulp difference of a middle value (other summation order) times its weight: bounded by 2^-11 sum |w x|
                got = _slice(det, int(op["out_buf"]), int(op["out_ld"]), int(op["out_coff"]), int(op["Cout"]), int(op["Ho"]), int(op["Wo"]), fr)
                got = got.float().cpu()
            elif wl["kind"] == "conv":
                keys = wl["wkey"] if isinstance(wl["wkey"], tuple) else (wl["wkey"],)
                ref = torch.cat([dt._conv_bn_act(x, sd, key, k, s_, pd, wl["act"], fp16=True, round_out=False) for key in keys], 1)
                absum = torch.cat([dt.conv_abs_sum(x, sd, key, s_, pd) for key in keys], 1)      # sum_k |w_k x_k| (+ |b|) per output
                got = _slice(det, int(op["out_buf"]), int(op["out_ld"]), int(op["out_coff"]), int(op["Cout"]), int(op["Ho"]), int(op["Wo"]), fr)
                got = got.float().cpu()
            else:                                                   # Detect 1x1 (models/yolo.py:46): fp16 weights, fp32 bias, fp32 output
                ref = F.conv2d(x, sd[wl["wkey"] + ".weight"].half().float(), sd[wl["wkey"] + ".bias"].float())
                absum = F.conv2d(x.abs(), sd[wl["wkey"] + ".weight"].half().float().abs(), sd[wl["wkey"] + ".bias"].float().abs())
                got = det.head_tensor(wl["level"], B_BENCH)[fr].cpu()
            ref, absum = ref.permute(0, 2, 3, 1), absum.permute(0, 2, 3, 1)
            err = (got - ref).abs()
            tol = 3e-4 + 6e-4 * ref.abs() + (2 * float(Cin * k * k) ** 0.5 * 2.0 ** -24 + extra_tol) * absum      # |SiLU'| <= 1.1: the pre-activation bound carries over
            bad = err > tol
            if bool(bad.any()):
                w_ = int(torch.argmax((err / tol).flatten()))
                detail = "got %.6g ref %.6g sum|wx| %.4g tol %.3g" % (float(got.flatten()[w_]), float(ref.flatten()[w_]), float(absum.flatten()[w_]), float(tol.flatten()[w_]))
            assert not bool(bad.any()), "op %d %s (%s, %dx%d %d->%d k%d s%d): %d values off, worst err/tol %.2f [%s]" % (
                oi, names[oi], wl["wkey"], H, W, Cin, int(op["Cout"]), k, s_, int(bad.sum()), float((err / tol).max()), detail)
            worst[names[oi]] = max(worst[names[oi]], float((err / tol).max()))
            n_conv += 1
        else:
            if int(op["type"]) == 1:
                ref = F.interpolate(x, scale_factor=2, mode="nearest")                     # nn.Upsample(None, 2, 'nearest')
            else:
                ref = F.max_pool2d(x, int(op["KH"]), int(op["stride"]), int(op["pad"]))  # SPPCSPC pools (cascaded), concat copies (k = 1)
            got = _slice(det, int(op["out_buf"]), int(op["out_ld"]), int(op["out_coff"]), Cin, ref.shape[2], ref.shape[3], fr).float().cpu()
            assert torch.equal(got, ref.permute(0, 2, 3, 1)), "op %d %s" % (oi, names[oi])
            n_other += 1
    assert ci == len(p.wlayout) and n_conv >= 95 and n_other + n_up >= 6 and n_up == 3      # w6: all three upsamples are read through
    print("per-op worst err / tol by kernel:", {k: "%.2e" % v for k, v in sorted(worst.items())})


def _imgs(frames_host, fr):
    return torch.from_numpy(frames_host[fr][..., ::-1].copy()).permute(0, 3, 1, 2).float() / 255.0      # tracker_dataloader.py:83-88


def test_heads_end_to_end_chaotic_weights_sanity(bench_det):
    """(b0) iid random weights: image -> raw Detect outputs of two frames vs the fp32 oracle.  Loose by nature (see the module docstring);
    the same numbers go into bench.py's JSON line (`parity`)."""
    from oracle import detector_torch as dt
    det, frames_host, out = bench_det
    fr = [0, B_BENCH - 1]
    raw = [r[fr].cpu() for r in out.raw()]
    _, ref32 = dt.forward(det.nodes, det._sd, _imgs(frames_host, fr), det.spec["anchors"])
    for l, (a, b) in enumerate(zip(raw, ref32)):
        scale, e32 = b.std().item(), (a - b).abs()
        print("chaotic weights, level %d: mean/max |err| = %.3e %.3e (logit std %.2f)" % (l, e32.mean().item(), e32.max().item(), scale))
        assert e32.mean().item() < 0.12 * scale and torch.isfinite(a).all(), l


def test_heads_end_to_end_against_fp32_oracle(smooth_det):
    """(b) well-conditioned weights: the raw head values of two full-size frames against the fp32 oracle's: mean |err| below 0.4 % of the
    logit spread (fp16 storage of ~60 tensors in a row without amplification: sqrt(60) * 2^-11 / sqrt(3) = 0.2 %), no value off by more
    than 4 %."""
    from oracle import detector_torch as dt
    det, frames_host, out = smooth_det
    fr = [0, B_BENCH - 1]
    raw = [r[fr].cpu() for r in out.raw()]
    _, ref32 = dt.forward(det.nodes, det._sd, _imgs(frames_host, fr), det.spec["anchors"])
    for l, (a, b) in enumerate(zip(raw, ref32)):
        scale, e32 = b.std().item(), (a - b).abs()
        print("level %d: mean/max |err| = %.3e %.3e (logit std %.2f)" % (l, e32.mean().item(), e32.max().item(), scale))
        assert 0.2 < scale < 20, (l, scale)                        # calibrated: logits O(1), the sigmoid is not saturated
        assert e32.mean().item() < 4e-3 * scale and e32.max().item() < 4e-2 * scale, l


def _kept_rows_check(det, frames_host, fr, coord_rel=0.0):
    """image -> final boxes, device vs fp32 oracle, by ANCHOR ROW (the device reports which candidate every kept detection is): rows kept on both sides must agree
    at SURVEY 8a's bar (same class, |dcoord| <= 1 px after round OR IoU >= 0.99 [OR, coord_rel > 0, |dcoord| <= 1 px + coord_rel x the box's larger side],
    |dconf| <= 5e-3); every row kept on ONE side only must be explained by a named greedy decision that flips within the measured score noise
    (oracle/detector_torch.py::explain_kept_set_difference: score tie with its rival, IoU at the NMS threshold, class tie of the rival, the max_det cut, conf_thres)."""
    from oracle import detector_torch as dt
    from tests import util
    out = det.forward(torch.from_numpy(frames_host).cuda(), fuse_decode=0.01)          # what bench.py times
    dets, nd = det.postprocess(out, 0.01, 0.45, None)
    torch.cuda.synchronize()
    det.check_overflow()
    keep = det.plan.post[out.pset].keep.cpu().numpy()
    cidx = det.candidate_arrays(out.pset)[3].cpu().numpy()
    dec, _ = dt.forward(det.nodes, det._sd, _imgs(frames_host, fr), det.spec["anchors"])
    stats = []
    for i, b in enumerate(fr):
        got, want = _device_candidates(det, out.pset, b), util.oracle_candidates(dec[i], 0.01)
        n = int(nd[b])
        kd = cidx[b][keep[b, :n]]                                   # device: kept anchor rows in output order
        kw = dt.nms_rows(want, 0.45)                                # oracle: utils/general.py:664-695 on its own candidates
        ref = dt.non_max_suppression(dec[i:i + 1], 0.01, 0.45)[0]
        assert len(ref) == len(kw) and np.array_equal(ref[:, 4].numpy(), np.array([want[r][1] for r in kw], np.float32))      # nms_rows IS non_max_suppression
        d = dets[b, :n].cpu().numpy()
        rb = dt.scale_coords_round((1280, 1280), ref[:, :4], (1280, 1280)).numpy()
        pos_d = {int(r): j for j, r in enumerate(kd)}
        both, worst_c, worst_s, n_iou, n_preclip = 0, 0.0, 0.0, 0, 0
        for j, r in enumerate(kw):
            if int(r) not in pos_d:
                continue
            both += 1
            row = d[pos_d[int(r)]]
            dc, ds = float(np.abs(row[:4] - rb[j]).max()), abs(float(row[4]) - float(ref[j, 4]))
            side = float(max(rb[j][2] - rb[j][0], rb[j][3] - rb[j][1]))
            ok_c = dc <= 1.0 or dt.box_iou_1(row[:4], rb[j]) >= 0.99 or dc <= 1.0 + coord_rel * side
            if not ok_c:
                # scale_coords' clip (utils/general.py:331-340) can cut a box that straddles the image border down to a fraction of itself: the same edge error is then
                # a larger share of what is left (a 781-px box clipped to 543 px: 2 px on one edge = IoU 0.9887).  The decode's own output -- the candidate before
                # clip / round -- is what 8a's bar is about there: it must pass (clip and round are exact integer operations on both sides)
                g, w_ = got[int(r)][0], want[int(r)][0]
                ok_c = (float(np.abs(g - w_).max()) <= 1.0 or dt.box_iou_1(g, w_) >= 0.99) and dc <= 3.0
                n_preclip += ok_c
            n_iou += dc > 1.0
            assert ok_c and ds <= 5e-3 and (row[5] == float(ref[j, 5]) or want[int(r)][3][int(row[5])] >= want[int(r)][1] - 5e-3), (b, int(r), row, rb[j], ref[j])
            worst_c, worst_s = max(worst_c, dc if dc <= 1.0 else 0.0), max(worst_s, ds)
        common = sorted(set(got) & set(want))
        noise = max(abs(got[r][1] - want[r][1]) for r in common)                 # the measured score noise of this frame's candidates ...
        assert noise <= SCORE_NOISE, (b, noise)                                  # ... must be inside the a-priori bound the explanations are allowed to use
        ex = dt.explain_kept_set_difference(got, kd, want, kw, score_noise=SCORE_NOISE)
        reasons = collections.Counter(v or "UNEXPLAINED" for v in ex.values())
        print("frame %d: oracle keeps %d, device %d, %d rows kept by both (all at the 8a bar: max |dcoord| %.0f px among the <= 1 px ones, %d pass on IoU >= 0.99 -- %d of them on the box before scale_coords' clip --, max |dconf| %.2e); "
              "%d rows kept on one side only: %s (measured score noise %.2e, allowed %.1e)" % (b, len(kw), n, both, worst_c, n_iou, n_preclip, worst_s, len(ex), dict(reasons), noise, SCORE_NOISE))
        assert len(kw) >= 100 and both >= 0.95 * len(kw)      # (measured: 290-298 of 300)
        if not all(v is not None for v in ex.values()):      # leave the two candidate sets behind for an off-line look (gpurun_out/ travels back from the GPU box)
            import pickle
            os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
            pickle.dump({"got": got, "want": want, "kd": kd, "kw": kw, "noise": noise, "ex": ex},
                        open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_debug_frame%d.pkl" % b), "wb"))
        assert all(v is not None for v in ex.values()), {k: v for k, v in ex.items() if v is None}
        stats.append((len(kw), n, both, dict(reasons)))
    return stats


def test_boxes_end_to_end_against_fp32_oracle(smooth_det):
    """(c) image -> (n, 6) rows [x1, y1, x2, y2, conf, cls]: device decode + NMS + scale_coords + round on the fp16 network's heads vs
    oracle/detector_torch (fp32 network, utils/general.py:607-695 NMS, general.py:319-340, track.py:240) on the same frames, all four Detect levels live.
    VERDICT r3 weak 2: no budget of unmatched boxes -- every kept row both sides share is at SURVEY 8a's bar, and every row only one side keeps is traced to
    the greedy decision that flipped and shown to be a tie within the frame's measured score noise."""
    det, frames_host, _ = smooth_det
    _kept_rows_check(det, frames_host, [0, B_BENCH - 1])


def _device_candidates(det, pset, b):
    """candidates of image b in post-processing set `pset` -> dict anchor row -> (xyxy, conf, cls)"""
    cbox, cscore, ccls, cidx, count = det.candidate_arrays(pset)
    n = int(count[b])
    rows = cidx[b, :n].cpu().numpy()
    assert len(set(rows.tolist())) == n                    # an anchor row is a candidate at most once
    return {int(r): (bx, float(s), int(c)) for r, bx, s, c in zip(rows, cbox[b, :n].cpu().numpy(), cscore[b, :n].cpu().numpy(), ccls[b, :n].cpu().numpy())}


def test_candidates_before_nms_against_fp32_oracle(smooth_det):
    """SURVEY 8a's box bar applied to the pre-NMS CANDIDATE set of the timed configuration (32 frames, 1280^2, the fused Detect
    epilogues of the benchmarked launch list), where no greedy decision can amplify a rounding difference:
      * every anchor row that is a candidate on both sides: same class (or one the fp32 oracle scores within 5e-3 of its best), |dcoord| <= 1 px,
        |dconf| <= 5e-3 -- ALL of them, not 97 %;
      * a row that is a candidate on one side only crossed conf_thres = 0.01 by rounding: its score is within 1e-3 of the threshold;
      * the NMS itself, run by the oracle's greedy restatement on the DEVICE's own candidates, keeps exactly the device's rows, in order.
    Together: whatever differs between the device's and the oracle's final boxes is fp16 noise in the scores flipping greedy NMS decisions
    (test_boxes_end_to_end_against_fp32_oracle counts those), not arithmetic of the decode / filter / NMS kernels."""
    from oracle import cnative, detector_torch as dt
    from tests import util
    det, frames_host, _ = smooth_det
    fr = [0, B_BENCH - 1]
    out = det.forward(torch.from_numpy(frames_host).cuda(), fuse_decode=0.01)          # what bench.py times: Detect decode + filter in the conv epilogue
    dets, nd = det.postprocess(out, 0.01, 0.45, None)
    torch.cuda.synchronize()
    det.check_overflow()
    ps = det.plan.post[out.pset]
    dec, _ = dt.forward(det.nodes, det._sd, _imgs(frames_host, fr), det.spec["anchors"])
    keep = ps.keep.cpu().numpy()
    for i, b in enumerate(fr):
        got = _device_candidates(det, out.pset, b)
        want = util.oracle_candidates(dec[i], 0.01)
        st = util.compare_candidate_sets(got, want, 0.01, px=1.0, dconf=5e-3, iou=0.99)      # 8a: IoU >= 0.99 / |dcoord| <= 1 px (the 400 ... 1700 px boxes of levels 2-3 pass on IoU)
        print("frame %d candidates:" % b, st)
        assert st["n_both"] >= 1000 and st["n_only_one_side"] <= 0.02 * st["n_both"], st
        assert st["frac_within_bar"] == 1.0 and st["out_of_coord_bar"] == [] and st["max_dconf"] <= 5e-3, st
        assert st["n_class_differs"] == 0 and st["max_margin_only_one_side"] <= 1e-3, st
        # greedy NMS (utils/general.py:664-695 + torchvision.ops.nms restated in oracle/y7t_oracle.c) on identical candidates: bit-exact keep list
        rows = np.array(sorted(got)); n = len(rows)
        cbox = np.stack([got[r][0] for r in rows]); cs = np.array([got[r][1] for r in rows], np.float32); cc = np.array([got[r][2] for r in rows], np.float32)
        order = np.lexsort((rows, -cs.astype(np.float64)))[:30000]
        k = cnative.nms((cbox + cc[:, None] * np.float32(4096)).astype(np.float32)[order], cs[order], 0.45)[:300]
        cidx = det.candidate_arrays(out.pset)[3][b].cpu().numpy()
        assert int(nd[b]) == len(k)
        np.testing.assert_array_equal(rows[order[k]], cidx[keep[b, :int(nd[b])]])


def test_all_levels_undamped_candidates_at_the_full_8a_bar(all_levels_det):
    """the configuration the other tests make easy, made hard -- all four Detect levels live (each supplies its quota of the ~2000 candidates, bench.LEVEL_QUOTA) and the
    random head's width / height logits UNDAMPED (std 2-3: the sigmoids saturate, boxes reach (2 s)^2 = 4 x the anchor: 2000+ px on levels 2-3, and ~0 px wide
    ones), 32 frames at 1280^2, fused Detect epilogues.  Against the fp32 oracle, for EVERY candidate both sides have (models/yolo.py:39-57, utils/general.py:629-662):
      * same class (or one the oracle scores within 5e-3 of its best) and |dconf| <= 5e-3 -- all of them;
      * coordinates at SURVEY 8a's full bar, IoU >= 0.99 OR |dcoord| <= 1 px: >= 98 % of them (the oracle's own fp16-storage emulation, scripts/parity_all_levels_cpu.py,
        says 98.9 %: a box edge moves by w (1 - s) 2 dt for a logit error dt, and dt -- fp16 storage of ~60 tensors, 0.13 % of the logit spread -- is what it is);
      * every candidate outside that bar is still tight -- |dcoord| <= 1 px + 0.75 % of its larger side (measured <= 0.61 %) -- and is a box no trained head emits: at least half the
        image side long (measured: 1200 ... 2450 px), or within 1.5 px (measured <= 1.27) with a side so short that this costs more than 1 % of IoU (23 x 190 px off by 1.26 px: 0.973).
    Candidates only one side has sit within 1e-3 of conf_thres."""
    from oracle import detector_torch as dt
    from tests import util
    det, frames_host, _ = all_levels_det
    fr = [0, B_BENCH - 1]
    out = det.forward(torch.from_numpy(frames_host).cuda(), fuse_decode=0.01)
    dets, nd = det.postprocess(out, 0.01, 0.45, None)
    torch.cuda.synchronize()
    det.check_overflow()
    dec, _ = dt.forward(det.nodes, det._sd, _imgs(frames_host, fr), det.spec["anchors"])
    rows0 = np.cumsum([0] + [3 * (1280 // s) ** 2 for s in (8, 16, 32, 64)])
    for i, b in enumerate(fr):
        got, want = _device_candidates(det, out.pset, b), util.oracle_candidates(dec[i], 0.01)
        st = util.compare_candidate_sets(got, want, 0.01, px=1.0, dconf=5e-3, iou=0.99)
        oob = st.pop("out_of_coord_bar")
        per_level = np.bincount(np.searchsorted(rows0, np.array(sorted(got)), side="right") - 1, minlength=4)[:4].tolist()
        print("frame %d, all levels undamped: candidates per level %s;" % (b, per_level), st, "| outside the coordinate bar:", oob[:6])
        assert st["n_both"] >= 1000 and st["n_only_one_side"] <= 0.02 * st["n_both"] and st["max_margin_only_one_side"] <= 1e-3, st
        assert all(v >= 50 for v in per_level) and per_level[2] + per_level[3] >= 100, per_level          # every level is really in play (bench.LEVEL_QUOTA)
        assert st["n_class_differs"] == 0 and st["max_dconf"] <= 5e-3, st
        assert st["frac_within_bar"] >= 0.98, st
        for o in oob:
            big, small = max(o["w"], o["h"]), min(o["w"], o["h"])
            assert o["dcoord"] <= (1.0 + 0.0075 * big if big >= 640 else 1.5), o      # (measured <= 0.61 % of the side over frames 0, 31, 39)      # (below half the image side: at most half a pixel past the bar -- on a short side that costs IoU: 23 x 190 px off by 1.26 px -> 0.973)


def test_all_levels_undamped_boxes_every_difference_explained(all_levels_det):
    """... and image -> final boxes in that configuration: rows both sides keep agree at the 8a bar (boxes larger than the image: 1 px + 0.5 % of the side), every row only
    one side keeps is traced to a greedy NMS decision tied within the measured score noise (see _kept_rows_check)."""
    det, frames_host, _ = all_levels_det
    _kept_rows_check(det, frames_host, [0, B_BENCH - 1], coord_rel=0.005)


def test_training_graph_checkpoint_on_the_device():
    """VERDICT r2 missing 2: the checkpoints the reference's training saves are cfg/training/yolov7-w6.yaml models -- IAuxDetect with ImplicitA /
    ImplicitM (models/yolo.py:111-158, common.py:433-456) and an aux branch that inference computes and discards (yaml :156-162).  A state dict of
    that shape (seeded, BatchNorm calibrated, non-trivial implicit layers, aux parameters present) through the product: (a) the launch list is the
    deploy graph's (aux branch dropped: 98 ops, the stride-2 layer + twin 1x1 pair being one), (b) every Detect level, teacher-forced on its actual fp16 input, equals im * (W (x + ia) + b) at
    the layer tolerance, (c) the raw heads end to end against the fp32 oracle (== the reference Model bit for bit on this graph,
    tests/test_detector_oracle.py) on well-conditioned weights."""
    from oracle import detector_torch as dt
    from tests import util
    from yolov7_tracker_amd.detector import arch, graph, model
    H, W, B = 384, 640, 2
    spec = arch.yolov7_w6_training(10)
    img = torch.rand((B, 3, H, W), generator=torch.Generator().manual_seed(21))
    plan0 = graph.lower(graph.parse(spec)[0], H, W, B)
    sd = util.training_checkpoint_state_dict(spec, plan0, seed=2, bn_bias_mean=2.0, calib_image=img[:1])
    det = model.Detector(spec, sd, img_size=(H, W), max_batch=B)
    dep = model.Detector(arch.yolov7_w6(10), None, img_size=(H, W), max_batch=B)
    n_fused = int((dep.plan.ops["korder"] == 11).sum())      # (the stride-2 layer + twin 1x1 pair is one op where the map is large enough for that kernel)
    assert len(det.plan.ops) == len(dep.plan.ops) == 99 - n_fused and det.launch_list(B) == dep.launch_list(B)
    out = det(img)[0]
    raw = [r.cpu() for r in out.raw()]
    p = det.plan
    for op in (p.ops[i] for i in p.detect_ops):
        l = int(op["detect_level"])
        x = det.buffer_view(int(op["in_buf"]), B, int(op["in_ld"])).view(B, int(op["H"]), int(op["W"]), -1)[..., int(op["in_coff"]):int(op["in_coff"]) + int(op["Cin"])]
        x = x.float().cpu().permute(0, 3, 1, 2)
        base = "model.%d" % next(n for n in det.nodes if n.kind == "detect").layer
        ia, im = sd["%s.ia.%d.implicit" % (base, l)], sd["%s.im.%d.implicit" % (base, l)]
        Wd, bd = sd["%s.m.%d.weight" % (base, l)], sd["%s.m.%d.bias" % (base, l)]
        ref = F.conv2d(x.double() + ia.double(), Wd.double(), bd.double()) * im.double()          # IAuxDetect.forward, yolo.py:136-137
        absum = F.conv2d(x.double().abs() + ia.double().abs(), Wd.double().abs(), bd.double().abs()) * im.double().abs()
        got = det.head_tensor(l, B).cpu().double().permute(0, 3, 1, 2)
        tol = 3e-4 + 2.0 ** -11 * absum        # the folded weights W * im are rounded to fp16 once (2^-11 relative each), fp32 accumulate
        assert bool(((got - ref).abs() <= tol).all()), (l, float(((got - ref).abs() / tol).max()))
    _, ref32 = dt.forward(det.nodes, sd, img, spec["anchors"])
    for l, (a, b) in enumerate(zip(raw, ref32)):
        scale, e = b.std().item(), (a - b).abs()
        print("training graph, level %d: mean/max |err| = %.3e %.3e (logit std %.2f)" % (l, e.mean().item(), e.max().item(), scale))
        assert e.mean().item() < 4e-3 * scale and e.max().item() < 4e-2 * scale, l
