"""Weights: deterministic random initialisation (no trained checkpoint ships with the reference), BatchNorm folding
(restating /root/reference/utils/torch_utils.py:181-201 fuse_conv_and_bn; BN eps = 1e-3 as set by
utils/torch_utils.py initialize_weights), IDetect implicit-layer folding (what tools/reparameterization.ipynb does:
m(x + ia) * im), and packing into the [Cout_pad][K_pad] fp16 layout of the implicit-GEMM kernel."""
import numpy as np

from .. import _lib
import torch

BN_EPS = 1e-3
GAIN_SILU, GAIN_LEAKY = 1.75, 1.40   # keep activation std ~O(1) through the depth of w6 / tiny


def random_state_dict(wlayout, seed=0, fused=False, bn_bias_mean=0.0):
    """Reference-style state dict ({wkey}.conv.weight / {wkey}.bn.* / model.N.m.L.{weight,bias}) with seeded values that
    keep activations O(1) through ~100 layers (so that fp16 storage is meaningful).

    bn_bias_mean: mean of the BatchNorm shifts beta (default 0: beta ~ N(0, 0.1)).  With beta ~ 0 every SiLU sees zero-mean unit-variance
    input and the random network sits on the chaotic side of the order/chaos transition: a perturbation grows relative to the signal by
    sqrt(E[silu'(z)^2] / Var[silu(z)]) = sqrt(0.38 / 0.31) ~ 1.1 per layer, ~300x over w6's depth, so fp16 rounding noise reaches 3-17 % of
    the head logits' spread.  beta ~ +2 moves the SiLUs into their near-linear region: the factor drops to ~1.0, the network keeps
    rich iid features with O(1) statistics, and fp16-vs-fp32 differences stay at the 0.2 % level -- the conditioning a trained detector
    has, which is what an end-to-end comparison against the fp32 oracle needs (tests/test_detector_pinned_gpu.py)."""
    import zlib
    sd = {}
    for w0 in wlayout:
      for w in ([dict(w0, wkey=k, cout=w0["cout"] // len(w0["wkey"])) for k in w0["wkey"]] if isinstance(w0["wkey"], tuple) else [w0]):
        cout, cin, k = w["cout"], w["cin"], w["k"]
        fan_in = cin * k * k
        rng = np.random.default_rng([seed, zlib.crc32(w["wkey"].encode())])   # per-layer stream: independent of plan order / fusion
        if w["kind"] != "conv":   # Detect 1x1 conv: plain conv with bias
            sd[w["wkey"] + ".weight"] = torch.from_numpy(rng.normal(0, 1.0 / np.sqrt(fan_in), (cout, cin, 1, 1)).astype(np.float32))
            sd[w["wkey"] + ".bias"] = torch.from_numpy(rng.normal(0, 0.5, cout).astype(np.float32))
            continue
        gain = {0: 1.0, 1: GAIN_SILU, 2: GAIN_LEAKY}[w.get('act', 1)]
        W = rng.normal(0, gain / np.sqrt(fan_in), (cout, cin, k, k)).astype(np.float32)
        g = rng.uniform(0.7, 1.3, cout).astype(np.float32)
        b = (rng.normal(0, 0.1, cout) + bn_bias_mean).astype(np.float32)
        mu = rng.normal(0, 0.1, cout).astype(np.float32)
        var = rng.uniform(0.8, 1.2, cout).astype(np.float32)
        if fused:
            scale = g / np.sqrt(var + BN_EPS)
            sd[w["wkey"] + ".conv.weight"] = torch.from_numpy(W * scale[:, None, None, None])
            sd[w["wkey"] + ".conv.bias"] = torch.from_numpy(b - mu * scale)
        else:
            sd[w["wkey"] + ".conv.weight"] = torch.from_numpy(W)
            sd[w["wkey"] + ".bn.weight"], sd[w["wkey"] + ".bn.bias"] = torch.from_numpy(g), torch.from_numpy(b)
            sd[w["wkey"] + ".bn.running_mean"], sd[w["wkey"] + ".bn.running_var"] = torch.from_numpy(mu), torch.from_numpy(var)
    return sd


@torch.no_grad()
def calibrate_bn(nodes, sd, hw=(640, 640), seed=0, image=None):
    """Data-dependent initialisation (host, once): set every BatchNorm's running statistics to the batch statistics its
    conv produces on a random low-resolution image (or on `image`, a (B,3,H,W) float tensor in [0,1]: statistics of the data the
    network will see, what a BN layer converges to in training), layer by layer, so the randomly initialised network keeps
    O(1) activations at any depth.  Weight INITIALISATION only; the hot path never runs through torch."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    vals = {0: torch.rand((1, 3) + tuple(hw), generator=g) if image is None else image.float()}
    for n in nodes[1:]:
        if n.kind == "detect" or any(j not in vals for j in n.src):
            continue
        if n.kind == "conv" and n.wkey + ".conv.weight" not in sd:      # a dead branch (the aux head of training graphs): not in the plan, no weights drawn
            continue
        x = vals[n.src[0]]
        if n.kind == "reorg":
            y = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
        elif n.kind == "conv":
            y = F.conv2d(x, sd[n.wkey + ".conv.weight"], None, stride=n.s, padding=n.p)
            mu, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
            sd[n.wkey + ".bn.running_mean"], sd[n.wkey + ".bn.running_var"] = mu.clone(), var.clone()
            y = (y - mu[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS) * sd[n.wkey + ".bn.weight"][None, :, None, None] \
                + sd[n.wkey + ".bn.bias"][None, :, None, None]
            y = F.silu(y) if n.act == 1 else (F.leaky_relu(y, 0.1) if n.act == 2 else y)
        elif n.kind == "concat":
            y = torch.cat([vals[j] for j in n.src], 1)
        elif n.kind == "up":
            y = F.interpolate(x, scale_factor=2, mode="nearest")
        elif n.kind == "pool":
            y = F.max_pool2d(x, n.k, n.s, n.p)
        else:
            raise NotImplementedError(n.kind)
        vals[n.idx] = y
    return sd


def folded(w, sd):
    """-> (W' float64 (cout, cin, k, k), b' float64 (cout,)) of one conv of the plan"""
    key = w["wkey"]
    if isinstance(key, tuple):   # fused twin 1x1 convs: stack the folded weights in channel order
        parts = [folded(dict(w, wkey=k), sd) for k in key]
        return np.concatenate([p[0] for p in parts], 0), np.concatenate([p[1] for p in parts], 0)
    if w["kind"] == "conv":
        W = sd[key + ".conv.weight"].detach().double().cpu().numpy()
        if key + ".bn.weight" in sd:
            g, b = sd[key + ".bn.weight"].double().cpu().numpy(), sd[key + ".bn.bias"].double().cpu().numpy()
            mu, var = sd[key + ".bn.running_mean"].double().cpu().numpy(), sd[key + ".bn.running_var"].double().cpu().numpy()
            scale = g / np.sqrt(var + BN_EPS)
            b0 = sd[key + ".conv.bias"].double().cpu().numpy() if key + ".conv.bias" in sd else 0.0
            return W * scale[:, None, None, None], (b0 - mu) * scale + b
        b0 = sd[key + ".conv.bias"].double().cpu().numpy() if key + ".conv.bias" in sd else np.zeros(W.shape[0])
        return W, b0
    W = sd[key + ".weight"].detach().double().cpu().numpy()
    b = sd[key + ".bias"].detach().double().cpu().numpy()
    if w["kind"] in ("IDetect", "IAuxDetect"):   # models/yolo.py:93-94 / :136-137: m(ia(x)) then im(.)
        base, lvl = key.rsplit(".m.", 1)
        ia, im = sd.get("%s.ia.%s.implicit" % (base, lvl)), sd.get("%s.im.%s.implicit" % (base, lvl))
        if ia is not None:
            b = b + W.reshape(W.shape[0], -1) @ ia.double().cpu().numpy().reshape(-1)
        if im is not None:
            s = im.double().cpu().numpy().reshape(-1)
            W, b = W * s[:, None, None, None], b * s
    return W, b


def panel_pack(blk, cin_pad, narrow=False):
    """[Cout_pad][K_pad] block with k = (kh*3 + kw)*Cin + ci  ->  the patch kernel's PANEL order (csrc/y7t_conv_patch.hip, korder 2):
    [n-tile of BN rows][K-step = 32-channel chunk * 9 + tap][row][four 16-byte slots], slot s of row r holding channel octet
    s ^ ((r >> 2) & 3) of the chunk -- byte for byte the image the kernel's buffer->LDS DMA leaves in LDS, so each K-step's
    panel is one contiguous run of full cache lines.  BN = 128 when Cout_pad allows, else 64."""
    cout_pad, K = blk.shape
    assert K == 9 * cin_pad and cin_pad % 64 == 0
    BN = 128 if cout_pad % 128 == 0 and not narrow else 64      # (narrow: korder 9)
    nc32 = cin_pad // 32
    a = blk.reshape(cout_pad // BN, BN, 9, nc32, 4, 8)           # [tile][row][tap][chunk][octet][8]
    a = a.transpose(0, 3, 2, 1, 4, 5)                            # [tile][chunk][tap][row][octet][8]
    r = np.arange(BN)
    src = np.arange(4)[None, :] ^ ((r[:, None] >> 2) & 3)        # slot s of row r <- octet s ^ ((r >> 2) & 3)
    a = np.take_along_axis(a, src[None, None, None, :, :, None], axis=4)
    return np.ascontiguousarray(a).reshape(cout_pad, K)


def panel_pack_linear(blk, narrow=False):
    """[Cout_pad][K_pad] block of a 1x1 layer -> panel order of the 32-deep generic kernel (csrc/y7t_conv.hip, korder 3):
    [n-tile of BN rows][K-step of 32 channels][row][four 16-byte slots], slot s of row r = channel octet s ^ ((r >> 2) & 3)."""
    cout_pad, K = blk.shape
    assert K % 32 == 0
    BN = 128 if cout_pad % 128 == 0 and not narrow else 64      # (narrow: korder 10)
    a = blk.reshape(cout_pad // BN, BN, K // 32, 4, 8).transpose(0, 2, 1, 3, 4)      # [tile][kstep][row][octet][8]
    r = np.arange(BN)
    src = np.arange(4)[None, :] ^ ((r[:, None] >> 2) & 3)
    a = np.take_along_axis(a, src[None, None, :, :, None], axis=3)
    return np.ascontiguousarray(a).reshape(cout_pad, K)


def panel_pack_p8(blk):
    """[Cout_pad][K] block of a 1x1 layer (Cout_pad % 256 == 0, K % 64 == 0) -> the panel order of csrc/y7t_conv_p8.hip (korder 7):
    [channel tile of 256][K-tile of 64][half h][row r of 128][eight 16-byte slots] = per (tile, K-tile) one contiguous 32 KiB block that IS the LDS image of the two
    channel half-tiles: row r of half h holds channel tile * 256 + (r // 32) * 64 + h * 32 + r % 32 (a wave's 64 output channels are contiguous), slot s of row r holds
    channel octet s ^ ((r >> 1) & 7) of the K-tile (the swizzle of the kernel's fragment reads)."""
    cout_pad, K = blk.shape
    assert cout_pad % 256 == 0 and K % 64 == 0
    a = blk.reshape(cout_pad // 256, 4, 2, 32, K // 64, 8, 8)            # [tile][wq][h][r % 32][ktile][octet][8]     channel = tile*256 + wq*64 + h*32 + r32
    a = a.transpose(0, 4, 2, 1, 3, 5, 6)                                  # [tile][ktile][h][wq][r32][octet][8]        row r = wq * 32 + r32
    a = np.ascontiguousarray(a).reshape(cout_pad // 256, K // 64, 2, 128, 8, 8)
    r = np.arange(128)
    src = np.arange(8)[None, :] ^ ((r[:, None] >> 1) & 7)
    a = np.take_along_axis(a, src[None, None, None, :, :, None], axis=4)
    return np.ascontiguousarray(a).reshape(cout_pad, K)


def s2_panel_width(cout_pad):
    """panel width of the stride-2 patch kernel for a layer (csrc/y7t_conv_patch_s2.hip::s2_bn, same rule)"""
    import os
    return 256 if cout_pad % 256 == 0 and _lib.switch("Y7T_CONV_PATCH_S2_BN", "") != "128" else 128


def panel_pack_s2(blk, cin_pad):
    """[Cout_pad][K_pad] block with k = (kh*3 + kw)*Cin + ci  ->  the STRIDE-2 patch kernel's panel order (csrc/y7t_conv_patch_s2.hip, korder 4):
    [n-tile of BN rows][K-step = 16-channel chunk * 9 + tap][row][two 16-byte slots], slot s of row r holding channel octet s ^ ((r >> 3) & 1)
    of the chunk -- the image the kernel's buffer->LDS DMA leaves in LDS.  BN = 256 when Cout_pad allows, else 128."""
    cout_pad, K = blk.shape
    assert K == 9 * cin_pad and cin_pad % 64 == 0 and cout_pad % 128 == 0
    BN = s2_panel_width(cout_pad)
    nc16 = cin_pad // 16
    a = blk.reshape(cout_pad // BN, BN, 9, nc16, 2, 8)           # [tile][row][tap][chunk][octet][8]
    a = a.transpose(0, 3, 2, 1, 4, 5)                            # [tile][chunk][tap][row][octet][8]
    r = np.arange(BN)
    src = np.arange(2)[None, :] ^ ((r[:, None] >> 3) & 1)        # slot s of row r <- octet s ^ ((r >> 3) & 1)
    a = np.take_along_axis(a, src[None, None, None, :, :, None], axis=4)
    return np.ascontiguousarray(a).reshape(cout_pad, K)


def pack_ws(blk):
    """[64][576] block of a 64 -> 64 3x3 layer with k = (kh*3 + kw)*64 + ci  ->  the register-fragment order of csrc/y7t_conv_ws.hip (korder 5): fragment
    f = (tap * 4 + ks) * 2 + i is 1 KiB = 64 lanes x 8 halves, lane l holding W[i*32 + l % 32][tap][ks*16 + 8*(l // 32) .. +7] -- the A operand of
    v_mfma_f32_32x32x16_f16 exactly as a lane keeps it, so that a wave loads a fragment with ONE 16-byte load per lane from consecutive addresses."""
    assert blk.shape == (64, 576)
    a = blk.reshape(2, 32, 9, 4, 2, 8)          # [i][l31][tap][ks][hi][8]
    a = a.transpose(2, 3, 0, 4, 1, 5)           # [tap][ks][i][hi][l31][8]   (lane = hi * 32 + l31)
    return np.ascontiguousarray(a).reshape(64, 576)


def pack_ws_s2(blk):
    """[128][576] block of the 64 -> 128 3x3 / stride 2 layer with k = (kh*3 + kw)*64 + ci  ->  the register-fragment order of csrc/y7t_conv_ws_s2.hip (korder 8):
    fragment f = (tap * 4 + ks) * 4 + q is 1 KiB = 64 lanes x 8 halves, lane l holding W[q*32 + l % 32][tap][ks*16 + 8*(l // 32) .. +7] (wave q keeps channels 32 q .. + 31)."""
    assert blk.shape == (128, 576)
    a = blk.reshape(4, 32, 9, 4, 2, 8)          # [q][l31][tap][ks][hi][8]
    a = a.transpose(2, 3, 0, 4, 1, 5)           # [tap][ks][q][hi][l31][8]   (lane = hi * 32 + l31)
    return np.ascontiguousarray(a).reshape(128, 576)


def pack_ws_s2_tail(blk):
    """[128][128] block of the twin 1x1 convolution fused behind the stride-2 layer (korder 11 ops; wlayout korder 12)  ->  its 8 x 4 A-fragments behind the 3x3 bank:
    fragment (ks, q) is 1 KiB, lane l holding W2[q*32 + l % 32][ks*16 + 8*(l // 32) .. +7]"""
    assert blk.shape == (128, 128)
    a = blk.reshape(4, 32, 8, 2, 8)             # [q][l31][ks][hi][8]
    a = a.transpose(2, 0, 3, 1, 4)              # [ks][q][hi][l31][8]
    return np.ascontiguousarray(a).reshape(128, 128)


def pack_ws128(blk):
    """[Cout_pad][1152] block of a 128 -> 128 k 3x3 layer (Cout_pad a multiple of 128) with k = (kh*3 + kw)*128 + ci  ->  the register-fragment order of
    csrc/y7t_conv_ws128.hip (korder 6): per 128-channel output tile n, fragment f = (n*72 + tap*8 + ks)*4 + q is 1 KiB = 64 lanes x 8 halves, lane l holding
    W[n*128 + q*32 + l % 32][tap][ks*16 + 8*(l // 32) .. +7]"""
    cp = blk.shape[0]
    assert blk.shape[1] == 1152 and cp % 128 == 0
    a = blk.reshape(cp // 128, 4, 32, 9, 8, 2, 8)      # [n][q][l31][tap][ks][hi][8]
    a = a.transpose(0, 3, 4, 1, 5, 2, 6)               # [n][tap][ks][q][hi][l31][8]   (lane = hi * 32 + l31)
    return np.ascontiguousarray(a).reshape(cp, 1152)


def pack(wlayout, sd, w_elems, b_elems):
    """-> (fp16 weight blob [w_elems], fp32 bias blob [b_elems]) in the kernel's [Cout_pad][K_pad] layout,
    k = (kh*KW + kw)*Cin_pad + ci"""
    wb = np.zeros(w_elems, np.float16)
    bb = np.zeros(b_elems, np.float32)
    for w in wlayout:
        W, b = folded(w, sd)
        cout, cin, k = w["cout"], w["cin"], w["k"]
        assert W.shape == (cout, cin, k, k), (w["wkey"], W.shape, (cout, cin, k, k))
        Wt = np.zeros((cout, k, k, w["cin_pad"]), np.float64)
        Wt[..., :cin] = W.transpose(0, 2, 3, 1)
        if w.get("korder") == 1:   # (cout, kh, kw, chunk, 64) -> (cout, kh, chunk, kw, 64)
            Wt = Wt.reshape(cout, k, k, w["cin_pad"] // 64, 64).transpose(0, 1, 3, 2, 4)
        blk = np.zeros((w["cout_pad"], w["K_pad"]), np.float16)
        blk[:cout, :w["K"]] = Wt.reshape(cout, -1).astype(np.float16)
        if w.get("korder") in (2, 9):
            blk = panel_pack(blk, w["cin_pad"], narrow=w["korder"] == 9)
        elif w.get("korder") in (3, 10):
            blk = panel_pack_linear(blk, narrow=w["korder"] == 10)
        elif w.get("korder") == 4:
            blk = panel_pack_s2(blk, w["cin_pad"])
        elif w.get("korder") == 5:
            blk = pack_ws(blk)
        elif w.get("korder") == 6:
            blk = pack_ws128(blk)
        elif w.get("korder") == 7:
            blk = panel_pack_p8(blk)
        elif w.get("korder") == 8:
            blk = pack_ws_s2(blk)
        elif w.get("korder") == 12:
            blk = pack_ws_s2_tail(blk)
        wb[w["w_off"]:w["w_off"] + blk.size] = blk.reshape(-1)
        bb[w["b_off"]:w["b_off"] + cout] = b.astype(np.float32)
    return wb, bb
