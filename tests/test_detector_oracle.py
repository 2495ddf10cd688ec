"""CPU: the torch-fp32 detector oracle (oracle/detector_torch.py) and the graph host logic, pinned against the
reference: (a) golden vectors recorded from the reference's models.yolo.Model / utils.general.non_max_suppression
(tests/golden/detector.npz), (b) the live reference where /root/reference exists."""
import os

import numpy as np
import pytest
import torch

from oracle import detector_torch as dt
from tests import util
from yolov7_tracker_amd.detector import arch, graph, weights

CASES = [("tiny", "yolov7-tiny", 80, (64, 96)), ("w6", "yolov7-w6", 10, (128, 128))]


def seeded_sd(name, nc, hw):
    spec = arch.ARCHS[name](nc)
    nodes, _ = graph.parse(spec)
    plan = graph.lower(graph.parse(spec)[0], hw[0], hw[1], 1)
    sd = weights.calibrate_bn(nodes, weights.random_state_dict(plan.wlayout, 0), hw=(128, 128), seed=0)
    return spec, nodes, plan, sd


@pytest.mark.parametrize("key,name,nc,hw", CASES)
def test_oracle_forward_and_nms_match_reference_golden(key, name, nc, hw):
    g = np.load(util.GOLDEN + "/detector.npz")
    spec, nodes, plan, sd = seeded_sd(name, nc, hw)
    chk = sum(float(v.double().sum()) for v in sd.values())
    if abs(chk - float(g[key + "_sd_checksum"])) > 1e-6 * abs(chk):
        pytest.skip("seeded weights differ on this host (different BLAS summation in the BN calibration)")
    dec, _ = dt.forward(nodes, sd, torch.from_numpy(g[key + "_img"]), spec["anchors"])
    np.testing.assert_allclose(dec.numpy(), g[key + "_decoded"], rtol=1e-4, atol=1e-4)
    nms = dt.non_max_suppression(torch.from_numpy(g[key + "_pred"]), 0.01, 0.45)[0]
    np.testing.assert_array_equal(nms.numpy(), g[key + "_nms"])


def test_graph_census_matches_survey():
    """w6 @ 1280, nc=10 -> 107 convs, 177.45 GMAC; tiny @ 640 nc=80 -> 58 convs, 6.85 GMAC"""
    p = graph.lower(graph.parse(arch.yolov7_w6(10))[0], 1280, 1280, 1)
    n_convs = lambda pl: sum(len(w["wkey"]) if isinstance(w["wkey"], tuple) else 1 for w in pl.wlayout)
    assert n_convs(p) == 107 and abs(p.macs / 1e9 - 177.45) < 0.01
    fused_s2 = 1      # ... and the stride-2 64 -> 128 layer with the first twin pair behind it (korder 11)
    assert (p.ops["type"] == 0).sum() == 107 - 11 - fused_s2       # the 11 twin 1x1 pairs of the ELAN blocks run as one launch each
    assert [h["stride"] for h in p.heads] == [8, 16, 32, 64] and sum(3 * h["ny"] * h["nx"] for h in p.heads) == 102000
    p = graph.lower(graph.parse(arch.yolov7_tiny(80))[0], 640, 640, 1)
    assert n_convs(p) == 58 and abs(p.macs / 1e9 - 6.85) < 0.01
    assert sum(3 * h["ny"] * h["nx"] for h in p.heads) == 25200


def test_concat_elimination_writes_disjoint_slices():
    p = graph.lower(graph.parse(arch.yolov7_w6(10))[0], 256, 256, 1)
    seen = {}
    for op in p.ops:
        key = int(op["out_buf"])
        rng = (int(op["out_coff"]), int(op["out_coff"]) + int(op["Cout"]))
        for a, b in seen.get(key, []):
            assert rng[1] <= a or rng[0] >= b, "overlapping writers in buffer %d" % key
        seen.setdefault(key, []).append(rng)
        assert rng[1] <= int(op["out_ld"])


def test_specs_equal_reference_yaml(have_reference):
    if not have_reference:
        pytest.skip("/root/reference not present")
    import yaml
    for fn, cfg in ((arch.yolov7_w6, "cfg/deploy/yolov7-w6.yaml"), (arch.yolov7_tiny, "cfg/deploy/yolov7-tiny.yaml")):
        y = yaml.safe_load(open("/root/reference/" + cfg))
        ref = y["backbone"] + y["head"]
        mine = fn(y["nc"])["layers"]
        norm = lambda L: [str(x).replace("'None'", "None") for x in L]
        assert norm(ref) == norm(mine) and y["anchors"] == fn(y["nc"])["anchors"]
        loaded = arch.load_yaml("/root/reference/" + cfg)
        assert norm(loaded["layers"]) == norm(ref)


def test_oracle_equals_live_reference_model(have_reference):
    if not have_reference:
        pytest.skip("/root/reference not present (golden vectors cover this)")
    from oracle import ref_harness
    spec, nodes, plan, sd = seeded_sd("yolov7-tiny", 80, (64, 96))
    m = ref_harness.build_reference_model("cfg/deploy/yolov7-tiny.yaml", 80)
    m.load_state_dict(sd, strict=False)
    img = torch.rand((2, 3, 64, 96), generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = m(img)
    dec, raw = dt.forward(nodes, sd, img, spec["anchors"])
    assert torch.equal(dec, ref[0])
    # BN folding of the product's packer == the reference's fuse()
    m.fuse()
    mods = dict(m.named_modules())
    for w in plan.wlayout[:8]:
        W, b = weights.folded(w, sd)
        keys = w["wkey"] if isinstance(w["wkey"], tuple) else (w["wkey"],)     # fused twin 1x1 convs: stacked in channel order
        Wr = np.concatenate([mods[k].conv.weight.detach().numpy() for k in keys], 0)
        br = np.concatenate([mods[k].conv.bias.detach().numpy() for k in keys], 0)
        np.testing.assert_allclose(W, Wr, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(b, br, rtol=1e-5, atol=1e-6)


def test_load_checkpoint_pickled_reference_model_with_autoanchor_buffers(have_reference, tmp_path):
    """attempt_load's host half on what the reference's training code saves: a PICKLED Model under 'model' (models/experimental.py:88-89),
    whose Detect anchor buffers were rewritten the way train.py's autoanchor does (utils/autoanchor.py:40-60: anchor_grid in pixels,
    anchors in stride units, model.yaml untouched).  The loaded spec must decode with the BUFFERS, and the oracle walk over the loaded
    graph + state dict must reproduce the reference model's own forward."""
    if not have_reference:
        pytest.skip("/root/reference not present")
    from oracle import ref_harness
    from yolov7_tracker_amd.detector import load_checkpoint
    m = ref_harness.build_reference_model("cfg/deploy/yolov7-tiny.yaml", 80)
    g = torch.Generator().manual_seed(5)
    for mod in m.modules():                                   # non-trivial BN statistics so the fold is exercised
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    det = m.model[-1]
    scale = torch.tensor([1.3, 0.8]).view(1, 1, 2)
    new_px = (det.anchors * det.stride.view(-1, 1, 1) * scale).round()
    det.anchors[:] = new_px / det.stride.view(-1, 1, 1)
    det.anchor_grid[:] = new_px.view(det.nl, 1, -1, 1, 1, 2)
    assert m.yaml["anchors"] == arch.TINY_ANCHORS                      # the yaml still says the old anchors
    img = torch.rand((1, 3, 64, 96), generator=g)
    with torch.no_grad():
        want = m(img)[0]
    path = str(tmp_path / "best.pt")
    with ref_harness.reference_pickle_context():
        torch.save({"model": m, "epoch": 12, "optimizer": None}, path)
        spec, sd, _ = load_checkpoint(path)
    assert spec["anchors"] == [[float(v) for v in lvl.reshape(-1)] for lvl in new_px] != arch.TINY_ANCHORS
    assert spec["nc"] == 80 and len(spec["layers"]) == len(arch.yolov7_tiny(80)["layers"])
    nodes, _ = graph.parse(spec)
    dec, _ = dt.forward(nodes, sd, img, spec["anchors"])
    assert torch.allclose(dec, want, rtol=1e-5, atol=1e-5)
    # a state-dict checkpoint that only carries `anchors` (stride units), with a yaml that gives an anchor COUNT (`anchors: 3`)
    sd2 = {k: v for k, v in sd.items() if not k.endswith("anchor_grid")}
    torch.save({"model": sd2}, str(tmp_path / "sd.pt"))
    import yaml as _yaml
    y = dict(m.yaml, anchors=3)
    cfg = str(tmp_path / "m.yaml")
    _yaml.safe_dump(y, open(cfg, "w"))
    spec2, _, _ = load_checkpoint(str(tmp_path / "sd.pt"), cfg=cfg)
    assert spec2["anchors"] == spec["anchors"]
    with pytest.raises(ValueError):
        load_checkpoint(_strip_anchor_buffers(sd2, tmp_path), cfg=cfg)


def _strip_anchor_buffers(sd, tmp_path):
    p = str(tmp_path / "noanchors.pt")
    torch.save({k: v for k, v in sd.items() if not k.endswith(".anchors")}, p)
    return p


def test_weight_panel_packings_are_permutations_with_the_documented_layout():
    """CPU: the two panel orders of detector/weights.py (korder 2: LDS-patch kernel; korder 3: 1x1 layers of the generic kernel)
    are pure permutations of the [Cout_pad][K] block, and element (tile, K-step, row, slot) is where the kernels' DMA expects it:
    slot s of row r holds channel octet s ^ ((r >> 2) & 3) of the step's 32 channels"""
    import numpy as np
    from yolov7_tracker_amd.detector import weights
    rng = np.random.default_rng(0)
    for cout_pad, cin in ((128, 64), (192, 128), (64, 192)):
        blk = rng.permutation(cout_pad * 9 * cin).astype(np.float64).reshape(cout_pad, 9 * cin)
        out = weights.panel_pack(blk, cin)
        assert out.shape == blk.shape and np.array_equal(np.sort(out.ravel()), np.sort(blk.ravel()))
        BN, nc32 = (128 if cout_pad % 128 == 0 else 64), cin // 32
        flat = out.ravel()
        for tile, c, tap, r, s in ((0, 0, 0, 0, 0), (cout_pad // BN - 1, nc32 - 1, 8, BN - 1, 3), (0, 1, 4, 37, 2), (0, nc32 - 1, 7, 21, 1)):
            o = ((((tile * nc32 + c) * 9 + tap) * BN + r) * 4 + s) * 8
            k = tap * cin + c * 32 + (s ^ ((r >> 2) & 3)) * 8
            assert np.array_equal(flat[o:o + 8], blk[tile * BN + r, k:k + 8])
    for cout_pad, K in ((128, 256), (192, 128), (64, 1024)):
        blk = rng.permutation(cout_pad * K).astype(np.float64).reshape(cout_pad, K)
        out = weights.panel_pack_linear(blk)
        assert np.array_equal(np.sort(out.ravel()), np.sort(blk.ravel()))
        BN, nk = (128 if cout_pad % 128 == 0 else 64), K // 32
        flat = out.ravel()
        for tile, st, r, s in ((0, 0, 0, 0), (cout_pad // BN - 1, nk - 1, BN - 1, 3), (0, 2, 37, 2)):
            o = (((tile * nk + st) * BN + r) * 4 + s) * 8
            k = st * 32 + (s ^ ((r >> 2) & 3)) * 8
            assert np.array_equal(flat[o:o + 8], blk[tile * BN + r, k:k + 8])


def test_stride2_panel_packing_and_opt_in_lowering(monkeypatch):
    """CPU: korder 4 (csrc/y7t_conv_patch_s2.hip, opt-in experiment): the packing is a permutation with slot s of row r = channel octet s ^ ((r >> 3) & 1) of the
    K-step's 16 channels; the default lowering of yolov7-w6 contains no korder-4 op, with Y7T_CONV_PATCH_S2=1 exactly the eight stride-2 3x3 layers with
    Cin % 64 == 0 carry it (the stem's successor 64 -> 128 included) and nothing else changes"""
    import numpy as np
    from yolov7_tracker_amd.detector import arch, graph, weights
    rng = np.random.default_rng(0)
    for cout_pad, cin in ((128, 64), (256, 128), (384, 64), (512, 192)):
        blk = rng.permutation(cout_pad * 9 * cin).astype(np.float64).reshape(cout_pad, 9 * cin)
        out = weights.panel_pack_s2(blk, cin)
        assert out.shape == blk.shape and np.array_equal(np.sort(out.ravel()), np.sort(blk.ravel()))
        BN, nc16 = (256 if cout_pad % 256 == 0 else 128), cin // 16
        flat = out.ravel()
        for tile, c, tap, r, s in ((0, 0, 0, 0, 0), (cout_pad // BN - 1, nc16 - 1, 8, BN - 1, 1), (0, 1, 4, 37, 1), (0, nc16 - 1, 7, 21, 0), (0, 2, 3, 8, 0)):
            o = ((((tile * nc16 + c) * 9 + tap) * BN + r) * 2 + s) * 8
            k = tap * cin + c * 16 + (s ^ ((r >> 3) & 1)) * 8
            assert np.array_equal(flat[o:o + 8], blk[tile * BN + r, k:k + 8])
    monkeypatch.setenv("Y7T_CONV_WS_S2", "0")      # (round 4: the 640^2 64 -> 128 layer otherwise goes to its own weights-stationary kernel, korder 8, before this rule is asked)
    monkeypatch.setenv("Y7T_CONV_WS128", "0")      # (round 5: the two Cin = 128 layers otherwise go to ws128's stride-2 form, korder 6)
    monkeypatch.setenv("Y7T_CONV_PATCH_S2", "0")
    base = graph.lower(graph.parse(arch.ARCHS["yolov7-w6"](10))[0], 1280, 1280, max_batch=32)
    assert not any(int(op["korder"]) == 4 for op in base.ops)
    monkeypatch.delenv("Y7T_CONV_PATCH_S2")
    # the default rule = where the kernel measured faster (round 3): 256-channel panels, Cin >= 128, >= 50 000 output pixels per launch
    auto = graph.lower(graph.parse(arch.ARCHS["yolov7-w6"](10))[0], 1280, 1280, max_batch=32)
    took = sorted((int(op["H"]), int(op["Cin"]), int(op["Cout"])) for op in auto.ops if int(op["korder"]) == 4)
    assert took == [(80, 512, 768), (160, 128, 256), (160, 256, 512), (320, 128, 256)], took
    assert not any(int(op["korder"]) == 4 for op in graph.lower(graph.parse(arch.ARCHS["yolov7-w6"](10))[0], 1280, 1280, max_batch=1).ops)      # batch 1: generic + split-K
    monkeypatch.setenv("Y7T_CONV_PATCH_S2", "1")
    exp = graph.lower(graph.parse(arch.ARCHS["yolov7-w6"](10))[0], 1280, 1280, max_batch=32)
    assert len(exp.ops) == len(base.ops)
    changed = [(a, b) for a, b in zip(base.ops, exp.ops) if int(a["korder"]) != int(b["korder"])]
    assert len(changed) == 8 and all(int(b["korder"]) == 4 and int(b["stride"]) == 2 and int(b["KH"]) == 3 and int(b["Cin"]) % 64 == 0 for _, b in changed)
    for a, b in zip(base.ops, exp.ops):
        for f in a.dtype.names:
            assert f == "korder" or a[f] == b[f]
    # Y7T_CONV_PATCH_S2_MIN_COUT is an EXPERIMENT switch: the product library ignores it; the lowering honours it only beside the measuring build (liby7t_ablate.so)
    monkeypatch.setenv("Y7T_CONV_PATCH_S2_MIN_COUT", "256")
    assert sum(int(op["korder"]) == 4 for op in graph.lower(graph.parse(arch.ARCHS["yolov7-w6"](10))[0], 1280, 1280, max_batch=32).ops) == 8
    monkeypatch.setenv("Y7T_LIB", "/somewhere/liby7t_ablate.so")
    wide = graph.lower(graph.parse(arch.ARCHS["yolov7-w6"](10))[0], 1280, 1280, max_batch=32)
    assert sum(int(op["korder"]) == 4 for op in wide.ops) == 7          # all but the 64 -> 128 layer at 640x640


def test_weights_stationary_128_packing_and_lowering(monkeypatch):
    """CPU: korder 6 = the filter bank of a 128 -> 128 k 3x3 layer as MFMA A-fragments per 128-channel output tile (csrc/y7t_conv_ws128.hip): a permutation with the
    documented index map; with Y7T_CONV_WS128=1 the lowering gives it the 128 -> 128 / 256 layers on maps of whole 4 x 16 tiles and nothing else changes; =0 leaves the
    plan without it"""
    from yolov7_tracker_amd.detector import arch, graph, weights
    blk = np.random.default_rng(1).permutation(256 * 1152).astype(np.float64).reshape(256, 1152)
    out = weights.pack_ws128(blk).ravel()
    assert np.array_equal(np.sort(out), np.sort(blk.ravel()))
    for n, tap, ks, q, lane in ((0, 0, 0, 0, 0), (1, 8, 7, 3, 63), (0, 4, 2, 1, 37), (1, 7, 5, 2, 5)):
        f = (n * 72 + tap * 8 + ks) * 4 + q
        assert np.array_equal(out[(f * 64 + lane) * 8:(f * 64 + lane) * 8 + 8], blk[n * 128 + q * 32 + lane % 32, tap * 128 + ks * 16 + 8 * (lane // 32):][:8])
    low = lambda: graph.lower(graph.parse(arch.ARCHS["yolov7-w6"](10))[0], 1280, 1280, max_batch=32)
    monkeypatch.setenv("Y7T_CONV_WS128", "0")
    base = low()
    assert not any(int(op["korder"]) == 6 for op in base.ops)
    monkeypatch.setenv("Y7T_CONV_WS128", "1")
    exp = low()
    took = [(int(op["H"]), int(op["Cin"]), int(op["Cout"]), int(op["stride"])) for op in exp.ops if int(op["korder"]) == 6]
    assert sorted(took) == sorted([(160, 128, 128, 1)] * 4 + [(80, 128, 128, 1)] * 6 + [(160, 128, 256, 1)] + [(320, 128, 256, 2), (160, 128, 256, 2)]), took      # (the last two: the stride-2 form)
    for a, b in zip(base.ops, exp.ops):
        for f in a.dtype.names:
            assert f == "korder" or a[f] == b[f] or (f in ("Cout_pad", "w_off", "bias_off") and int(b["korder"]) == 6), (f, a[f], b[f])
    monkeypatch.setenv("Y7T_LIB", "/somewhere/liby7t_ablate.so")      # beside the measuring build the stride-2 form alone can be switched off (an experiment switch)
    monkeypatch.setenv("Y7T_CONV_WS128_S2", "0")
    assert sum(int(op["korder"]) == 6 for op in low().ops) == 11


def test_weights_stationary_packing_and_lowering(monkeypatch):
    """CPU: korder 5 = the 64 x 576 filter bank of a 64 -> 64 3x3 layer as MFMA A-fragments (csrc/y7t_conv_ws.hip): a permutation with the documented
    index map; the lowering sends exactly the seven 64 -> 64 layers on the 320^2 / 160^2 maps to it and nothing else changes"""
    from yolov7_tracker_amd.detector import arch, graph, weights
    blk = np.random.default_rng(0).permutation(64 * 576).astype(np.float64).reshape(64, 576)
    out = weights.pack_ws(blk).ravel()
    assert np.array_equal(np.sort(out), np.sort(blk.ravel()))
    for tap, ks, i, lane in ((0, 0, 0, 0), (8, 3, 1, 63), (4, 2, 0, 37), (7, 1, 1, 5)):
        f = (tap * 4 + ks) * 2 + i
        assert np.array_equal(out[(f * 64 + lane) * 8:(f * 64 + lane) * 8 + 8], blk[i * 32 + lane % 32, tap * 64 + ks * 16 + 8 * (lane // 32):][:8])
    low = lambda: graph.lower(graph.parse(arch.ARCHS["yolov7-w6"](10))[0], 1280, 1280, max_batch=32)
    monkeypatch.setenv("Y7T_CONV_WS", "0")
    base = low()
    assert not any(int(op["korder"]) == 5 for op in base.ops)
    monkeypatch.delenv("Y7T_CONV_WS")
    exp = low()
    took = [(int(op["H"]), int(op["Cin"]), int(op["Cout"])) for op in exp.ops if int(op["korder"]) == 5]
    assert took == [(320, 64, 64)] * 4 + [(160, 64, 64)] * 3
    for a, b in zip(base.ops, exp.ops):
        for f in a.dtype.names:
            assert f == "korder" or a[f] == b[f]
    assert not any(int(op["korder"]) == 5 for op in graph.lower(graph.parse(arch.ARCHS["yolov7-w6"](10))[0], 1280, 1280, max_batch=1).ops)      # 400 tiles: too few


def test_patch_eligibility_rule():
    """the Python mirror of the dispatcher's rule for the LDS-patch kernel (detector/graph.py::patch_eligible)"""
    from yolov7_tracker_amd.detector import graph
    ok = lambda H, W, ci, co, B, **kw: graph.patch_eligible(H, W, ci, co, kw.get("k", 3), kw.get("s", 1), kw.get("p", 1), co, 0, 0, B)
    assert ok(80, 80, 256, 256, 32) and ok(320, 320, 64, 64, 32) and ok(40, 40, 384, 384, 32) and ok(20, 20, 512, 1024, 32)
    assert ok(20, 20, 512, 512, 32) and ok(20, 20, 256, 256, 32)      # round 4: 64-row panels (korder 9) where 128 rows give fewer than 256 workgroups, threshold 200
    assert graph.patch_panel_rows(20, 20, 512, 32) == 64 and graph.patch_panel_rows(20, 20, 1024, 80) == 128 and graph.patch_panel_rows(80, 80, 192, 32) == 64
    assert graph.patch_panel_rows(20, 20, 512, 80) == 64 and graph.patch_panel_rows(20, 20, 1024, 32) == 64      # round 6, measured at 80 frames: 64-row panels below 512 tiles of 128 rows (500 for the 20 x 20 512 -> 512 layers)
    assert ok(20, 20, 512, 512, 8) and not ok(20, 20, 512, 512, 4)      # 100 workgroups' worth of 256-pixel tiles (64-row panels): 8 * 400 * 8; 4 frames are too few
    assert ok(80, 80, 256, 256, 1) and not ok(80, 80, 256, 128, 1)      # batch 1 (round 4, profiles/r04_latency_lowering.txt): 100 workgroups win against split-K, 50 with Cin = 256 lose
    assert ok(80, 80, 128, 128, 1) and not ok(40, 40, 128, 128, 1)      # ... 50 are enough when K is short (Cin <= 128)
    assert not ok(80, 80, 256, 256, 32, s=2) and not ok(80, 80, 256, 256, 32, k=1, p=0)
    assert not ok(80, 80, 96, 256, 32)                 # Cin % 64
    assert not ok(24, 24, 256, 256, 64)                # 24x24: 16x16 tiles 56 %, 32x8 tiles 75 %, no strip tiling for this width


def test_lowering_of_the_benchmarked_list_by_weight_order(monkeypatch):
    """Which kernel family each conv of the benchmarked configuration (w6 @ 1280, 80 frames per forward since round 6; 40 in round 5, 32 before) is lowered to, as `korder` counts -- every rule behind them was set by
    an in-session A/B on the device (profiles/r03_*, r04_*); a change here is a change of the measured launch list.  And the batch-1 list."""
    for k in ("Y7T_CONV_WS128", "Y7T_LIB", "Y7T_CONV_WS_S2_FUSE", "Y7T_CONV_PATCH_MIN_PIX", "Y7T_CONV_PATCH_PANEL64_BELOW", "Y7T_CONV_1X1_PANEL64_BELOW", "Y7T_CONV_P8", "Y7T_CONV_WS", "Y7T_CONV_WS_S2",
              "Y7T_CONV_PATCH_S2", "Y7T_CONV_PATCH", "Y7T_CONV_VARIANT", "Y7T_CONV_WPANEL"):
        monkeypatch.delenv(k, raising=False)
    import collections
    hist = lambda B: dict(sorted(collections.Counter(int(o["korder"]) for o in graph.lower(graph.parse(arch.yolov7_w6(10))[0], 1280, 1280, B).ops if int(o["type"]) == 0).items()))
    # 0 stem (fused frame -> conv kernel); 1 generic 3x3 (three stride-2 layers); 2 LDS-patch (16x16 tiles + 40-wide strips); 3 1x1 panels (incl. 3 upsample-on-read, 4 Detect);
    # 4 stride-2 LDS-patch; 5 weights-stationary 64 -> 64; 6 weights-stationary 128 -> 128 k (round 5: eleven of the former korder-2 launches at stride 1 + the two Cin = 128 stride-2 layers, formerly korder 4); 7 p8; 9 patch with 64-row panels (the 20x20 layers); 10 1x1 with 64-row panels (< 500 tiles);
    # 11 the stride-2 weights-stationary layer + the twin 1x1 behind it in one launch
    # (round 6: p8 on every 1x1 grid of >= 1500 tiles or of one full round of the chip, 64-row strip panels below 512 tiles: profiles/r06_batch_80.txt)
    assert hist(80) == {0: 1, 1: 3, 2: 22, 3: 13, 4: 2, 5: 7, 6: 13, 7: 23, 9: 10, 11: 1}      # the benchmarked list
    assert hist(32) == {0: 1, 1: 3, 2: 21, 3: 19, 4: 2, 5: 7, 6: 13, 7: 10, 9: 11, 10: 7, 11: 1}
    assert hist(40) == {0: 1, 1: 3, 2: 21, 3: 18, 4: 2, 5: 7, 6: 13, 7: 17, 9: 11, 10: 1, 11: 1}      # 40 frames: six of the 20x20 1x1 layers reach 500 tiles of 128 rows
    h1 = hist(1)
    assert h1 == {0: 1, 1: 33, 2: 8, 3: 8, 9: 16, 10: 28, 11: 1} and h1.get(5, 0) == 0 and h1.get(7, 0) == 0 and h1.get(4, 0) == 0      # one frame: no persistent 64 -> 64 / p8 / stride-2 patch launches
    monkeypatch.setenv("Y7T_CONV_WS_S2_FUSE", "0")      # an experiment switch: nothing changes for the product library ...
    assert hist(32)[11] == 1
    monkeypatch.setenv("Y7T_LIB", "/somewhere/liby7t_ablate.so")      # ... beside the measuring build the fusion can be switched off
    assert hist(32)[8] == 1 and 11 not in hist(32) and hist(32)[3] == 20


def test_training_graph_spec_and_liveness():
    """cfg/training/yolov7-w6.yaml (what the reference's training saves, README.md:101): the aux branch -- convs 118-121 and IAuxDetect's m2 convs,
    computed and discarded at inference (models/yolo.py:141-153) -- never reaches the launch list; the main head folds ImplicitA / ImplicitM"""
    dep = graph.lower(graph.parse(arch.yolov7_w6(10))[0], 1280, 1280, 1)
    trn = graph.lower(graph.parse(arch.yolov7_w6_training(10))[0], 1280, 1280, 1)
    assert len(trn.ops) == len(dep.ops) == 98 and abs(trn.macs - dep.macs) < 1 and trn.arena_bytes == dep.arena_bytes
    assert [w["kind"] for w in trn.wlayout if w["kind"] != "conv"] == ["IAuxDetect"] * 4
    sd = util.training_checkpoint_state_dict(arch.yolov7_w6_training(10), graph.lower(graph.parse(arch.yolov7_w6_training(10))[0], 128, 128, 1))
    w = next(w for w in trn.wlayout if w["kind"] == "IAuxDetect")
    W, b = weights.folded(w, sd)
    W0, b0 = sd[w["wkey"] + ".weight"].double().numpy(), sd[w["wkey"] + ".bias"].double().numpy()
    base, lvl = w["wkey"].rsplit(".m.", 1)
    ia, im = sd["%s.ia.%s.implicit" % (base, lvl)].double().numpy().reshape(-1), sd["%s.im.%s.implicit" % (base, lvl)].double().numpy().reshape(-1)
    np.testing.assert_allclose(W, W0 * im[:, None, None, None], rtol=1e-12)
    np.testing.assert_allclose(b, (b0 + W0.reshape(len(b0), -1) @ ia) * im, rtol=1e-12)         # im * (W (x + ia) + b)


def test_training_spec_equals_reference_yaml_and_oracle_equals_reference_model(have_reference):
    if not have_reference:
        pytest.skip("/root/reference not present")
    import yaml
    from oracle import ref_harness
    y = yaml.safe_load(open("/root/reference/cfg/training/yolov7-w6.yaml"))
    norm = lambda L: [str(x).replace("'None'", "None") for x in L]
    assert norm(y["backbone"] + y["head"]) == norm(arch.yolov7_w6_training(y["nc"])["layers"])
    spec = arch.yolov7_w6_training(10)
    nodes, _ = graph.parse(spec)
    plan = graph.lower(graph.parse(spec)[0], 64, 64, 1)
    sd = util.training_checkpoint_state_dict(spec, plan)
    m = ref_harness.build_reference_model("cfg/training/yolov7-w6.yaml", 10)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected[:5]
    assert all(".anchor" in k or "num_batches_tracked" in k for k in missing), [k for k in missing if ".anchor" not in k][:5]
    img = torch.rand((1, 3, 64, 128), generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        ref = m(img)
    dec, raw = dt.forward(nodes, sd, img, spec["anchors"])
    assert torch.equal(dec, ref[0])                                   # IAuxDetect.forward inference branch: main levels only (yolo.py:141-158)


def test_kept_set_difference_explainer_and_full_coordinate_bar():
    """the checkers the GPU box-level tests rely on (oracle/detector_torch.py: nms_rows, compare_candidate_sets(iou=...), explain_kept_set_difference), exercised where no
    GPU exists: the oracle's fp16-storage emulation stands in for the device on a 512 x 512 frame, all four Detect levels live.  nms_rows must BE
    non_max_suppression (utils/general.py:607-695); every kept-set difference between the two precisions must get a named reason; a difference that rounding
    noise cannot produce (a candidate whose score was moved by 100 x the noise) must NOT."""
    from yolov7_tracker_amd import synth
    torch.set_num_threads(8)
    nc, H = 10, 512
    spec = arch.ARCHS["yolov7-w6"](nc)
    frames = synth.make_frames(1, 80, H, seq_idx=0)
    img = (torch.from_numpy(frames[:1][..., ::-1].copy()).permute(0, 3, 1, 2).float() / 255.0).contiguous()
    nodes, _ = graph.parse(spec)
    plan = graph.lower(graph.parse(spec)[0], H, H, 1)
    sd = weights.calibrate_bn(nodes, weights.random_state_dict(plan.wlayout, 0, bn_bias_mean=2.0), seed=0, image=img)
    na, no = 3, nc + 5
    for k in list(sd):
        if ".m." in k and k.endswith(".weight"):
            w = sd[k].clone().view(na, no, -1)
            w[:, 2:4] *= 0.25
            sd[k] = w.view(na * no, -1, 1, 1)
    _, raw16 = dt.forward(nodes, sd, img, spec["anchors"], fp16=True)
    logits = torch.cat([r[0, ..., 4].reshape(-1) for r in raw16])
    shift = float(np.log(0.01 / 0.99)) - torch.quantile(logits.float(), 1.0 - 1500 / logits.numel()).item()      # detector/model.py::plant_objectness_bias
    base = "model.%d" % next(n for n in nodes if n.kind == "detect").layer
    for l in range(4):
        b = sd["%s.m.%d.bias" % (base, l)].float().clone().view(na, no)
        b[:, 4] += shift
        b[:, 5:] += 4.0
        sd["%s.m.%d.bias" % (base, l)] = b.view(-1)
    dec32, _ = dt.forward(nodes, sd, img, spec["anchors"])
    dec16, _ = dt.forward(nodes, sd, img, spec["anchors"], fp16=True)
    got, want = dt.candidates(dec16[0], 0.01), dt.candidates(dec32[0], 0.01)
    st = dt.compare_candidate_sets(got, want, 0.01, px=1.0, dconf=5e-3, iou=0.99)
    assert st["n_both"] > 1000 and st["frac_within_bar"] == 1.0 and st["out_of_coord_bar"] == [] and st["n_class_differs"] == 0, st
    assert dt.compare_candidate_sets(got, want, 0.01, px=1e-3, dconf=5e-3, iou=0.999999)["frac_within_bar"] < 1.0      # the bar is a bar
    kg, kw = dt.nms_rows(got), dt.nms_rows(want)
    ref = dt.non_max_suppression(dec32, 0.01, 0.45)[0]
    assert len(ref) == len(kw) and np.array_equal(ref[:, 4].numpy(), np.array([want[r][1] for r in kw], np.float32))
    assert np.allclose(ref[:, :4].numpy(), np.stack([want[r][0] for r in kw]))
    noise = max(abs(got[r][1] - want[r][1]) for r in set(got) & set(want))
    ex = dt.explain_kept_set_difference(got, kg, want, kw, score_noise=noise)
    assert set(ex) == set(kg.tolist()) ^ set(kw.tolist())
    assert all(v is not None for v in ex.values()), ex
    # a genuine difference (a kernel that drops a detection nothing overlaps, ranked far from the max_det cut) must stay unexplained
    lonely = next(int(r) for r in kw[:50] if r in set(kg.tolist()) and not any(
        o != r and got[o][2] == got[r][2] and dt.box_iou_1(got[o][0], got[r][0]) > 0.3 for o in kg))
    kb = np.array([r for r in kg if r != lonely])
    exb = dt.explain_kept_set_difference(got, kb, want, kw, score_noise=noise)
    assert lonely in exb and exb[lonely] is None, exb
    assert dt.box_iou_1([0, 0, 10, 10], [0, 0, 10, 5]) == 0.5 and dt.box_iou_1([0, 0, 1, 1], [2, 2, 3, 3]) == 0.0
