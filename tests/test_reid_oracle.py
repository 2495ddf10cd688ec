"""CPU: the ReID oracle (oracle/reid_torch.py) pinned against the reference's own OSNet class -- seeded random weights and the checkpoint
the reference ships (weights/osnet_x0_25.pth) -- and the host lowering of the product (tracker/reid.py) checked for consistency."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import reid_torch
from yolov7_tracker_amd.tracker import reid

REF = "/root/reference"


def _reference_osnet():
    spec = importlib.util.spec_from_file_location("ref_osnet", os.path.join(REF, "tracker/reid_models/OSNet.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.osnet_x0_25(num_classes=1, pretrained=False).eval()


def test_oracle_equals_reference_osnet(have_reference):
    if not have_reference:
        pytest.skip("/root/reference not present")
    net = _reference_osnet()
    x = torch.randn((3, 3, 128, 64), generator=torch.Generator().manual_seed(0))
    sd = reid.random_state_dict(reid.osnet_spec(0.25), 1)
    r = net.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all(k.startswith("classifier") for k in r.missing_keys)      # same parameter names as torchreid's OSNet
    with torch.no_grad():
        assert torch.equal(reid_torch.osnet_forward(sd, x), net(x))
    real = torch.load(os.path.join(REF, "weights/osnet_x0_25.pth"), map_location="cpu")
    real = {k.replace("module.", "", 1): v for k, v in real.get("state_dict", real).items()}
    net.load_state_dict({k: v for k, v in real.items() if not k.startswith("classifier")}, strict=False)
    with torch.no_grad():
        want = net(x)
    assert torch.equal(reid_torch.osnet_forward(real, x), want) and float(want.abs().mean()) > 0.1


def _reference_deepsort_net():
    """the reference's Net class (reid_models/deepsort_reid.py); the module imports cv2 / torchvision for its Extractor, absent here: stubbed"""
    import sys
    import types
    stubs = {}
    for name in ("cv2", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            stubs[name] = sys.modules[name] = types.ModuleType(name)
    if "torchvision" in stubs:
        stubs["torchvision"].transforms = sys.modules["torchvision.transforms"]
    try:
        spec = importlib.util.spec_from_file_location("ref_deepsort_reid", os.path.join(REF, "tracker/reid_models/deepsort_reid.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for name in stubs:
            del sys.modules[name]
    return m.Net(reid=True).eval()


def test_oracle_equals_reference_deepsort_net(have_reference):
    """oracle/reid_torch.py::deepsort_net_forward == the reference's own Net(reid=True) on the same `net_dict`, bit for bit"""
    if not have_reference:
        pytest.skip("/root/reference not present")
    net = _reference_deepsort_net()
    sd = reid.deepsort_net_random_state_dict(2)
    r = net.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all(k.startswith("classifier") or k.endswith("num_batches_tracked") for k in r.missing_keys)
    x = torch.randn((3, 3, 128, 64), generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        want = net(x)
    got = reid_torch.deepsort_net_forward(sd, x)
    assert tuple(want.shape) == (3, 512) and torch.equal(got, want)
    np.testing.assert_allclose(got.norm(dim=1).numpy(), 1.0, rtol=1e-6)


def _interpret_ops(ops, bufs, w, x):
    """run an op list of tracker/reid.py (include/y7t.h: y7t_reid_op) in float32 torch ops with the semantics of csrc/y7t_reid.hip's kernels --
    checks the host lowering (BatchNorm folding, weight layouts, buffer wiring) on the CPU.  x: (N, 3, H, W) -> (N, feat)"""
    import torch.nn.functional as F
    w = torch.from_numpy(np.asarray(w))
    N = x.shape[0]
    val = {0: x.permute(0, 2, 3, 1).contiguous()}            # NHWC like the arena
    for o in ops:
        t, H, W, C, Ho, Wo, Co, k = (int(o[f]) for f in ("type", "H", "W", "C", "Ho", "Wo", "Co", "k"))
        xin = val[int(o["in_buf"])]
        nchw = lambda a, h, ww, c: a.reshape(N, h, ww, c).permute(0, 3, 1, 2)
        bias = w[int(o["b_off"]):int(o["b_off"]) + Co] if int(o["b_off"]) >= 0 else None
        if t == reid.CONV:
            wt = w[int(o["w_off"]):int(o["w_off"]) + Co * k * k * C]
            wt = wt.reshape(k, k, C, Co).permute(3, 2, 0, 1) if int(o["w_kmajor"]) else wt.reshape(Co, k, k, C).permute(0, 3, 1, 2)
            y = F.conv2d(nchw(xin, H, W, C), wt, bias, int(o["s"]), int(o["p"]))
            y = F.relu(y) if int(o["relu"]) else y
        elif t == reid.DWCONV3:
            wt = w[int(o["w_off"]):int(o["w_off"]) + C * 9].reshape(C, 1, 3, 3)
            y = F.conv2d(nchw(xin, H, W, C), wt, w[int(o["b_off"]):int(o["b_off"]) + C], 1, 1, groups=C)
            y = F.relu(y) if int(o["relu"]) else y
        elif t == reid.MAXPOOL3S2:
            y = F.max_pool2d(nchw(xin, H, W, C), 3, 2, padding=1)
        elif t == reid.AVGPOOL2:
            y = F.avg_pool2d(nchw(xin, H, W, C), 2)
        elif t == reid.GATE_ACC:
            R = int(o["R"])
            a = nchw(xin, H, W, C)
            w1 = w[int(o["w_off"]):int(o["w_off"]) + R * C].reshape(R, C)
            b1 = w[int(o["b_off"]):int(o["b_off"]) + R]
            w2 = w[int(o["w2_off"]):int(o["w2_off"]) + C * R].reshape(C, R)
            b2 = w[int(o["b2_off"]):int(o["b2_off"]) + C]
            g = torch.sigmoid(F.relu(a.mean((2, 3)) @ w1.T + b1) @ w2.T + b2)
            y = a * g[:, :, None, None]
            if not int(o["relu"]):                               # relu == 1 marks the first branch (overwrite)
                y = y + val[int(o["out_buf"])].permute(0, 3, 1, 2)
        elif t == reid.ADD_RELU:
            y = F.relu(nchw(xin, H, W, C) + nchw(val[int(o["aux_buf"])], H, W, C))
        elif t == reid.GAP:
            val[int(o["out_buf"])] = nchw(xin, H, W, C).mean((2, 3))
            continue
        elif t == reid.FC:
            y = xin.reshape(N, C) @ w[int(o["w_off"]):int(o["w_off"]) + Co * C].reshape(Co, C).T + bias
            val[int(o["out_buf"])] = F.relu(y) if int(o["relu"]) else y
            continue
        elif t == reid.L2NORM:
            v = xin.reshape(N, C)
            val[int(o["out_buf"])] = v / v.norm(dim=1, keepdim=True)
            continue
        # the fp16 op types: values are kept as float32 tensors that hold fp16-representable numbers (one rounding per stored tensor)
        elif t == reid.H_PACK:
            y = torch.zeros((N, H, W, 16))
            y[..., :3] = xin.reshape(N, H, W, 3)
            val[int(o["out_buf"])] = y.half().float()
            continue
        elif t == reid.H_CONV:
            K = k * k * C
            Kp = (K + 63) // 64 * 64
            blk = w[int(o["w_off"]):int(o["w_off"]) + Co * Kp // 2].numpy().view(np.float16).reshape(Co, Kp)
            assert not blk[:, K:].any()
            wt = torch.from_numpy(blk[:, :K].astype(np.float32)).reshape(Co, k, k, C).permute(0, 3, 1, 2)
            y = F.conv2d(nchw(xin, H, W, C), wt, bias, int(o["s"]), int(o["p"])).half().float()
        elif t == reid.H_MAXPOOL_RELU:
            y = F.max_pool2d(F.relu(nchw(xin, H, W, C)), 3, 2, padding=1)
        elif t == reid.H_RELU:
            y = F.relu(nchw(xin, H, W, C))
        elif t == reid.H_ADD_RELU:
            y = F.relu(nchw(xin, H, W, C) + nchw(val[int(o["aux_buf"])], H, W, C)).half().float()
        elif t == reid.H_GAP_L2NORM:
            v = nchw(xin, H, W, C).mean((2, 3))
            val[int(o["out_buf"])] = v / v.norm(dim=1, keepdim=True)
            continue
        else:
            raise AssertionError(t)
        assert int(bufs[int(o["out_buf"])]) == y.shape[1] * y.shape[2] * y.shape[3] // (2 if t >= reid.H_PACK else 1)
        val[int(o["out_buf"])] = y.permute(0, 2, 3, 1).contiguous()
    return val[int(ops[-1]["out_buf"])]


def test_op_lists_encode_their_networks():
    """both lowerings -- OSNet x0_25 and the reference's DeepSORT Net -- interpreted on the CPU equal the oracle networks (fp32, BatchNorm folded:
    1e-4 of the output scale)"""
    x = torch.randn((2, 3, 128, 64), generator=torch.Generator().manual_seed(4))
    spec = reid.osnet_spec(0.25)
    sd = reid.random_state_dict(spec, 5)
    got, want = _interpret_ops(*reid.lower(sd, spec), x), reid_torch.osnet_forward(sd, x)
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    sd = reid.deepsort_net_random_state_dict(6)
    ops, bufs, w = reid.lower_deepsort_net(sd)
    got, want = _interpret_ops(ops, bufs, w, x), reid_torch.deepsort_net_forward(sd, x)
    assert tuple(got.shape) == (2, 512) and float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert w.size == sum(v.numel() for k, v in sd.items() if k.endswith(".weight") and v.dim() == 4) + sum(v.numel() for k, v in sd.items() if k.endswith("running_mean"))
    assert int(ops[-1]["type"]) == reid.L2NORM and sum(int(o["type"]) == reid.ADD_RELU for o in ops) == 8
    with pytest.raises(ValueError):
        reid.lower_deepsort_net(sd, 256, 128)
    # the MFMA op list of the same network: fp16 weights and stored activations (one rounding per tensor), fp32 accumulation -- what the
    # detector's conv kernels compute; shortcut projections as centre-tap 3x3 filters
    ops, bufs, w = reid.lower_deepsort_net_f16(sd)
    got = _interpret_ops(ops, bufs, w, x)
    err = float((got - want).abs().max()) / float(want.abs().max())
    cos = float(((got * want).sum(1) / (got.norm(dim=1) * want.norm(dim=1))).min())
    assert err <= 5e-3 and cos >= 1 - 1e-5, (err, cos)
    assert sum(int(o["type"]) == reid.H_CONV for o in ops) == 1 + 16 + 3 and all(int(o["k"]) == 3 for o in ops if int(o["type"]) == reid.H_CONV)
    for o in ops:        # alignment the conv launcher needs: 16-byte aligned weights / bias, channel counts
        if int(o["type"]) == reid.H_CONV:
            assert int(o["w_off"]) % 4 == 0 and int(o["b_off"]) % 4 == 0 and int(o["C"]) % 8 == 0 and int(o["Co"]) % 64 == 0
    print("fp16 op list vs fp32 oracle: max err %.2e of the feature scale, min cosine 1 - %.1e" % (err, 1 - cos))


def test_lowering_covers_every_parameter():
    spec = reid.osnet_spec(0.25)
    sd = reid.random_state_dict(spec, 0)
    ops, bufs, w = reid.lower(sd, spec)
    assert len(bufs) > 100 and bufs[0] == 128 * 64 * 3 and bufs[-1] == 512
    n_conv_w = sum(v.numel() for k, v in sd.items() if k.endswith(".weight") and v.dim() >= 2)
    n_bn = sum(v.numel() for k, v in sd.items() if k.endswith("running_mean"))
    n_gate_b = sum(v.numel() for k, v in sd.items() if ".gate." in k and k.endswith(".bias"))
    n_plain_1x1 = sum(1 for o in ops if int(o["type"]) == reid.CONV and int(o["b_off"]) < 0)
    assert w.size == n_conv_w + n_bn + n_gate_b      # every conv / linear weight once, one folded bias per BatchNorm, the gate biases
    assert n_plain_1x1 == 6 * 10                       # the linear 1x1 of every LightConv3x3: 10 per OSBlock
    assert sum(int(o["type"]) == reid.GATE_ACC for o in ops) == 24
    # buffers are written before they are read
    written = {0}
    for o in ops:
        assert int(o["in_buf"]) in written and (int(o["type"]) != reid.ADD_RELU or int(o["aux_buf"]) in written)
        written.add(int(o["out_buf"]))


def test_preprocess_oracle_shapes():
    frame = np.random.default_rng(0).integers(0, 256, (200, 300, 3)).astype(np.uint8)
    x = reid_torch.preprocess(frame, [[10, 20, 60, 150], [100.7, 0.2, 130.9, 64.5]])
    assert tuple(x.shape) == (2, 3, 128, 64) and torch.isfinite(x).all()
    # an exact 128 x 64 crop is only normalised
    y = reid_torch.preprocess(frame, [[5, 7, 69, 135]])[0]
    want = ((frame[7:135, 5:69].astype(np.float32) / 255 - np.float32([0.485, 0.456, 0.406])) / np.float32([0.229, 0.224, 0.225])).transpose(2, 0, 1)
    np.testing.assert_allclose(y.numpy(), want, rtol=1e-6, atol=1e-6)


def _interpret_fused_blob(blob, x):
    """run OSNet x0_25 FROM the parameter blob of the fused kernel (tracker/reid.py::pack_fused), reading it in the kernel's consumption order
    (csrc/y7t_reid_fused.hip) -- in float32 torch ops, so what is checked is the blob's content and layout, not the kernel's fp16 storage.
    x: (N, 3, 128, 64) -> (N, 512)"""
    import torch.nn.functional as F
    buf = memoryview(blob.tobytes())
    pos = [0]

    def take(nbytes, dtype):
        a = np.frombuffer(buf[pos[0]:pos[0] + nbytes], dtype=dtype)
        pos[0] += nbytes
        return a

    def unfrag(ng, nk):
        fr = take(ng * nk * 512, np.float16).reshape(ng, nk, 64, 4).astype(np.float32)
        M = np.zeros((ng * 16, nk * 16), np.float32)
        lane = np.arange(64)
        for g in range(ng):
            for k in range(nk):
                for e in range(4):
                    M[g * 16 + lane % 16, k * 16 + 4 * (lane // 16) + e] = fr[g, k, :, e]
        return torch.from_numpy(M)

    def f32(n):
        return torch.from_numpy(take(4 * n, np.float32).copy())

    def conv1x1(t, cout_p, cin_p, bias=True, relu=False):
        W = unfrag(cout_p // 16, cin_p // 16)
        b = f32(cout_p) if bias else None
        y = F.conv2d(t, W[:, :, None, None], b)
        return F.relu(y) if relu else y

    # conv1 7x7: fragments [kh*2 + half][lane][e]: row lane % 16, input pixel kw = 4 * half + lane // 16, channel e
    fr = take(14 * 512, np.float16).reshape(14, 64, 4).astype(np.float32)
    W1 = np.zeros((16, 3, 7, 7), np.float32)
    lane = np.arange(64)
    for kh in range(7):
        for h in range(2):
            kw = 4 * h + lane // 16
            for c in range(3):
                ok = kw < 7
                W1[(lane % 16)[ok], c, kh, kw[ok]] = fr[kh * 2 + h, ok, c]
    t = F.relu(F.conv2d(x, torch.from_numpy(W1), f32(16), stride=2, padding=3))
    t = F.max_pool2d(t, 3, 2, 1)

    def block(t, cin, cout, midp, R):
        x1_W = unfrag(midp // 16, cin // 16); x1_b = f32(midp)
        w1 = f32(R * midp).reshape(R, midp); b1 = f32(4)[:R]; w2 = f32(R * midp).reshape(R, midp); b2 = f32(midp)
        W3 = unfrag(cout // 16, midp // 16); b3 = f32(cout)
        Wd = unfrag(cout // 16, cin // 16) if cin != cout else None
        x1 = F.relu(F.conv2d(t, x1_W[:, :, None, None], x1_b))
        x2 = 0
        for n in (1, 2, 3, 4):
            u = x1
            for _ in range(n):
                Wl = unfrag(midp // 16, midp // 16)
                dw = f32(midp * 9).reshape(midp // 8, 9, 8).permute(0, 2, 1).reshape(midp, 1, 3, 3)
                db = f32(midp)
                u = F.relu(F.conv2d(F.conv2d(u, Wl[:, :, None, None]), dw, db, padding=1, groups=midp))
            g = u.mean((2, 3))
            g = torch.sigmoid(F.relu(g @ w1.T + b1) @ w2 + b2)
            x2 = x2 + u * g[:, :, None, None]
        y = F.conv2d(x2, W3[:, :, None, None], b3)
        return F.relu(y + (F.conv2d(t, Wd[:, :, None, None]) if Wd is not None else t))

    t = block(t, 16, 64, 16, 1); t = block(t, 64, 64, 16, 1)
    t = F.avg_pool2d(conv1x1(t, 64, 64, relu=True), 2)
    t = block(t, 64, 96, 32, 1); t = block(t, 96, 96, 32, 1)
    t = F.avg_pool2d(conv1x1(t, 96, 96, relu=True), 2)
    t = block(t, 96, 128, 32, 2); t = block(t, 128, 128, 32, 2)
    t = conv1x1(t, 128, 128, relu=True)
    v = t.mean((2, 3))
    wt = torch.from_numpy(take(64 * 512 * 4, np.float16).reshape(64, 512, 2).astype(np.float32))      # [channel pair][output][2]
    Wf = wt.permute(1, 0, 2).reshape(512, 128)
    out = F.relu(v @ Wf.T + f32(512))
    assert pos[0] == len(buf), "the kernel's walk and the blob's length disagree"
    return out


def test_fused_blob_encodes_the_network():
    """tracker/reid.py::pack_fused (BatchNorm folded, MFMA fragment order, 24 -> 32 channel padding of stage 3, downsample bias merged into conv3's)
    read back in the order csrc/y7t_reid_fused.hip consumes it == the oracle network, up to the fp16 rounding of the stored weights"""
    spec = reid.osnet_spec(0.25)
    sd = reid.random_state_dict(spec, 5)
    blob = reid.pack_fused(sd, spec)
    x = torch.randn((3, 3, 128, 64), generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        got = _interpret_fused_blob(blob, x)
        want = reid_torch.osnet_forward(sd, x)
    assert got.shape == want.shape == (3, 512) and float(want.abs().mean()) > 0.05
    assert float((got - want).abs().max()) <= 2e-3 * float(want.abs().max())


def test_identity_feature_scene_is_deterministic():
    """synth.make_identity_features (the scenes of the deepsort_identity* goldens): same arguments -> same detections and the same feature per box"""
    from yolov7_tracker_amd import synth
    d1, f1 = synth.make_identity_features(12, 30, 640, seq_idx=3, dim=64, miss=0.2)
    d2, f2 = synth.make_identity_features(12, 30, 640, seq_idx=3, dim=64, miss=0.2)
    assert all(np.array_equal(a, b) for a, b in zip(d1, d2))
    for d in d1:
        a, b = f1(d[:, :4]), f2(d[:, :4])
        assert a.shape == (len(d), 64) and np.array_equal(a, b)
        if len(d):
            np.testing.assert_allclose(np.linalg.norm(a, axis=1), 1.0, rtol=1e-5)
