"""ReID embedding for DeepSORT on the device: OSNet (x0_25 by default) over crops taken from the frame in HBM.

Host half (this file): lower the OSNet graph of /root/reference/tracker/reid_models/OSNet.py:282-438 (osnet_x0_25 :567-579:
channels [16, 64, 96, 128], two OSBlocks per stage, fc 512) into the op list of include/y7t.h (`y7t_reid_op`), fold every BatchNorm
into its convolution / linear layer, pack the fp32 weights into one blob.  Device half: csrc/y7t_reid.hip.  `ReIDExtractor` is what
DeepSORT's `get_feature` seam (deepsort.py:19-41) calls: `features_for_boxes(frame, tlbrs)` does crop + /255 + resize to 128 x 64 +
Normalize on the GPU like the reference's Extractor (reid_models/deepsort_reid.py:112-153) and returns (N, 512) device features.

Weights: a state dict with torchreid's OSNet parameter names (what weights/osnet_x0_25.pth holds) or seeded random ones."""
import ctypes

import numpy as np
import torch

from .. import _lib

OP_DTYPE = np.dtype([("type", "<i4"), ("in_buf", "<i4"), ("out_buf", "<i4"), ("aux_buf", "<i4"), ("H", "<i4"), ("W", "<i4"), ("C", "<i4"),
                     ("Ho", "<i4"), ("Wo", "<i4"), ("Co", "<i4"), ("k", "<i4"), ("s", "<i4"), ("p", "<i4"), ("relu", "<i4"), ("R", "<i4"),
                     ("w_kmajor", "<i4"), ("w_off", "<i8"), ("b_off", "<i8"), ("w2_off", "<i8"), ("b2_off", "<i8")], align=False)
assert OP_DTYPE.itemsize == 96      # == sizeof(y7t_reid_op)
CONV, DWCONV3, MAXPOOL3S2, AVGPOOL2, GATE_ACC, ADD_RELU, GAP, FC, L2NORM, H_PACK, H_CONV, H_MAXPOOL_RELU, H_RELU, H_ADD_RELU, H_GAP_L2NORM = range(15)
BN_EPS = 1e-5


def osnet_spec(width=0.25, feature_dim=512):
    """channels of osnet_x1_0 .. osnet_x0_25 (OSNet.py:522-579)"""
    ch = {1.0: [64, 256, 384, 512], 0.75: [48, 192, 288, 384], 0.5: [32, 128, 192, 256], 0.25: [16, 64, 96, 128]}[width]
    return {"channels": ch, "layers": [2, 2, 2], "feature_dim": feature_dim}


def random_state_dict(spec, seed=0):
    """seeded OSNet state dict with torchreid's names (Kaiming-style scales, non-trivial BatchNorm statistics)"""
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(name, co, ci, k, groups=1, bias=False):
        fan = ci // groups * k * k
        sd[name + ".weight"] = torch.from_numpy(rng.normal(0, (1.0 / fan) ** 0.5, (co, ci // groups, k, k)).astype(np.float32))
        if bias:
            sd[name + ".bias"] = torch.from_numpy(rng.normal(0, 0.1, co).astype(np.float32))

    def bn(name, c):
        sd[name + ".weight"] = torch.from_numpy(rng.uniform(0.7, 1.3, c).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32))
        sd[name + ".running_mean"] = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32))
        sd[name + ".running_var"] = torch.from_numpy(rng.uniform(0.6, 1.4, c).astype(np.float32))

    def light(name, c):
        conv(name + ".conv1", c, c, 1)
        conv(name + ".conv2", c, c, 3, groups=c)
        bn(name + ".bn", c)
    ch = spec["channels"]
    conv("conv1.conv", ch[0], 3, 7)
    bn("conv1.bn", ch[0])
    for si, (cin, cout, nblk) in enumerate(zip(ch[:-1], ch[1:], spec["layers"])):
        stage = "conv%d" % (si + 2)
        for bi in range(nblk):
            b = "%s.%d" % (stage, bi)
            i_c = cin if bi == 0 else cout
            mid = cout // 4
            conv(b + ".conv1.conv", mid, i_c, 1)
            bn(b + ".conv1.bn", mid)
            light(b + ".conv2a", mid)
            for tag, n in (("conv2b", 2), ("conv2c", 3), ("conv2d", 4)):
                for j in range(n):
                    light("%s.%s.%d" % (b, tag, j), mid)
            conv(b + ".gate.fc1", mid // 16, mid, 1, bias=True)
            conv(b + ".gate.fc2", mid, mid // 16, 1, bias=True)
            conv(b + ".conv3.conv", cout, mid, 1)
            bn(b + ".conv3.bn", cout)
            if i_c != cout:
                conv(b + ".downsample.conv", cout, i_c, 1)
                bn(b + ".downsample.bn", cout)
        if si < 2:      # reduce_spatial_size: Conv1x1 + AvgPool2d(2)
            conv("%s.%d.0.conv" % (stage, nblk), cout, cout, 1)
            bn("%s.%d.0.bn" % (stage, nblk), cout)
    conv("conv5.conv", ch[3], ch[3], 1)
    bn("conv5.bn", ch[3])
    fd = spec["feature_dim"]
    sd["fc.0.weight"] = torch.from_numpy(rng.normal(0, (1.0 / ch[3]) ** 0.5, (fd, ch[3])).astype(np.float32))
    sd["fc.0.bias"] = torch.from_numpy(rng.normal(0, 0.1, fd).astype(np.float32))
    bn("fc.1", fd)
    return sd


class _Lowering:
    def __init__(self, sd, in_h, in_w):
        self.sd, self.ops, self.bufs, self.w = sd, [], [], []
        self.w_floats = 0
        self.in_buf = self.buf(in_h * in_w * 3)

    def buf(self, floats_per_crop):
        self.bufs.append(int(floats_per_crop))
        return len(self.bufs) - 1

    def put(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        off = self.w_floats
        self.w.append(arr)
        self.w_floats += arr.size
        return off

    def _bn(self, name):
        g, b = self.sd[name + ".weight"].double().numpy(), self.sd[name + ".bias"].double().numpy()
        mu, var = self.sd[name + ".running_mean"].double().numpy(), self.sd[name + ".running_var"].double().numpy()
        scale = g / np.sqrt(var + BN_EPS)
        return scale, b - mu * scale

    def op(self, **kw):
        o = np.zeros((), OP_DTYPE)
        o["aux_buf"], o["b_off"] = -1, -1
        for k, v in kw.items():
            o[k] = v
        self.ops.append(o)

    def conv(self, x, H, W, ci, co, k, s, p, wname, bnname=None, relu=0, kmajor=False):
        Wt = self.sd[wname + ".weight"].double().numpy()                       # (co, ci, k, k)
        bias = np.zeros(co)
        if bnname is not None:
            scale, bias = self._bn(bnname)
            Wt = Wt * scale[:, None, None, None]
            if wname + ".bias" in self.sd:                                     # a biased conv in front of the BatchNorm
                bias = bias + scale * self.sd[wname + ".bias"].double().numpy()
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        y = self.buf(Ho * Wo * co)
        # (co, kh, kw, ci): a thread walks its own filter; kmajor (kh, kw, ci, co): neighbouring threads (output channels) read neighbouring
        # weights -- the layout for wide layers.  Same products in the same order either way
        self.op(type=CONV, in_buf=x, out_buf=y, H=H, W=W, C=ci, Ho=Ho, Wo=Wo, Co=co, k=k, s=s, p=p, relu=relu, w_kmajor=int(kmajor),
                w_off=self.put(Wt.transpose(2, 3, 1, 0) if kmajor else Wt.transpose(0, 2, 3, 1)), b_off=self.put(bias) if bnname is not None else -1)
        return y, Ho, Wo

    def light(self, x, H, W, c, name):
        """LightConv3x3 (OSNet.py:128-157): 1x1 linear, depthwise 3x3, BN, ReLU"""
        y, _, _ = self.conv(x, H, W, c, c, 1, 1, 0, name + ".conv1")
        scale, bias = self._bn(name + ".bn")
        Wd = self.sd[name + ".conv2.weight"].double().numpy().reshape(c, 9) * scale[:, None]
        z = self.buf(H * W * c)
        self.op(type=DWCONV3, in_buf=y, out_buf=z, H=H, W=W, C=c, Ho=H, Wo=W, Co=c, k=3, s=1, p=1, relu=1, w_off=self.put(Wd), b_off=self.put(bias))
        return z

    def osblock(self, x, H, W, cin, cout, name):
        mid = cout // 4
        x1, _, _ = self.conv(x, H, W, cin, mid, 1, 1, 0, name + ".conv1.conv", name + ".conv1.bn", relu=1)
        acc = self.buf(H * W * mid)
        scratch = self.buf(2 * mid)                                            # per crop: pooled vector + gates
        g = name + ".gate"
        R = mid // 16
        w1, b1 = self.put(self.sd[g + ".fc1.weight"].numpy().reshape(R, mid)), self.put(self.sd[g + ".fc1.bias"].numpy())
        w2, b2 = self.put(self.sd[g + ".fc2.weight"].numpy().reshape(mid, R)), self.put(self.sd[g + ".fc2.bias"].numpy())
        for bi, (tag, n) in enumerate((("conv2a", 1), ("conv2b", 2), ("conv2c", 3), ("conv2d", 4))):
            t = x1
            for j in range(n):
                t = self.light(t, H, W, mid, "%s.%s" % (name, tag) if tag == "conv2a" else "%s.%s.%d" % (name, tag, j))
            self.op(type=GATE_ACC, in_buf=t, out_buf=acc, aux_buf=scratch, H=H, W=W, C=mid, R=R, relu=int(bi == 0), w_off=w1, b_off=b1, w2_off=w2, b2_off=b2)
        x3, _, _ = self.conv(acc, H, W, mid, cout, 1, 1, 0, name + ".conv3.conv", name + ".conv3.bn")
        idn = x
        if cin != cout:
            idn, _, _ = self.conv(x, H, W, cin, cout, 1, 1, 0, name + ".downsample.conv", name + ".downsample.bn")
        y = self.buf(H * W * cout)
        self.op(type=ADD_RELU, in_buf=x3, aux_buf=idn, out_buf=y, H=H, W=W, C=cout)
        return y


def lower(sd, spec, in_h=128, in_w=64):
    """-> (ops structured array, buffer sizes in floats per crop, weight blob float32)"""
    L = _Lowering(sd, in_h, in_w)
    ch = spec["channels"]
    x, H, W = L.conv(L.in_buf, in_h, in_w, 3, ch[0], 7, 2, 3, "conv1.conv", "conv1.bn", relu=1)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = L.buf(Ho * Wo * ch[0])
    L.op(type=MAXPOOL3S2, in_buf=x, out_buf=y, H=H, W=W, C=ch[0], Ho=Ho, Wo=Wo, Co=ch[0])
    x, H, W = y, Ho, Wo
    for si, (cin, cout, nblk) in enumerate(zip(ch[:-1], ch[1:], spec["layers"])):
        stage = "conv%d" % (si + 2)
        for bi in range(nblk):
            x = L.osblock(x, H, W, cin if bi == 0 else cout, cout, "%s.%d" % (stage, bi))
        if si < 2:
            x, _, _ = L.conv(x, H, W, cout, cout, 1, 1, 0, "%s.%d.0.conv" % (stage, nblk), "%s.%d.0.bn" % (stage, nblk), relu=1)
            y = L.buf((H // 2) * (W // 2) * cout)
            L.op(type=AVGPOOL2, in_buf=x, out_buf=y, H=H, W=W, C=cout, Ho=H // 2, Wo=W // 2, Co=cout)
            x, H, W = y, H // 2, W // 2
    x, _, _ = L.conv(x, H, W, ch[3], ch[3], 1, 1, 0, "conv5.conv", "conv5.bn", relu=1)
    v = L.buf(ch[3])
    L.op(type=GAP, in_buf=x, out_buf=v, H=H, W=W, C=ch[3])
    fd = spec["feature_dim"]
    scale, bias = L._bn("fc.1")
    Wf = sd["fc.0.weight"].double().numpy() * scale[:, None]
    bf = sd["fc.0.bias"].double().numpy() * scale + bias
    out = L.buf(fd)
    L.op(type=FC, in_buf=v, out_buf=out, H=1, W=1, C=ch[3], Co=fd, relu=1, w_off=L.put(Wf), b_off=L.put(bf))
    return np.array(L.ops, dtype=OP_DTYPE), L.bufs, np.concatenate(L.w)


def macs_per_crop(ops):
    """multiply-accumulates of one crop through an op list: (dense on the matrix cores, everything else) -- the algorithmic work the cfg4 bench line prices the
    fused OSNet kernel with (OSNet x0_25 at 128 x 64: 1x1 convs + the 7x7 stem + fc on MFMA, depthwise 3x3 / gates / pools on the VALU)"""
    dense = other = 0
    for o in ops:
        t = int(o["type"])
        if t == CONV:
            dense += int(o["Ho"]) * int(o["Wo"]) * int(o["Co"]) * int(o["k"]) ** 2 * int(o["C"])
        elif t == FC:
            dense += int(o["C"]) * int(o["Co"])
        elif t == DWCONV3:
            other += int(o["Ho"]) * int(o["Wo"]) * int(o["C"]) * 9
        elif t == GATE_ACC:
            other += int(o["H"]) * int(o["W"]) * int(o["C"]) * 2 + 2 * int(o["C"]) * int(o["R"])
    return dense, other


def deepsort_net_random_state_dict(seed=0):
    """seeded `net_dict` of the reference's DeepSORT embedding network (reid_models/deepsort_reid.py:62-110; the classifier is not part of the
    reid=True forward and is left out)"""
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(name, co, ci, k, bias=False):
        sd[name + ".weight"] = torch.from_numpy(rng.normal(0, (2.0 / (ci * k * k)) ** 0.5, (co, ci, k, k)).astype(np.float32))
        if bias:
            sd[name + ".bias"] = torch.from_numpy(rng.normal(0, 0.1, co).astype(np.float32))

    def bn(name, c):
        sd[name + ".weight"] = torch.from_numpy(rng.uniform(0.7, 1.3, c).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32))
        sd[name + ".running_mean"] = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32))
        sd[name + ".running_var"] = torch.from_numpy(rng.uniform(0.6, 1.4, c).astype(np.float32))
    conv("conv.0", 64, 3, 3, bias=True)
    bn("conv.1", 64)
    cin = 64
    for li, cout in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for bi in range(2):
            b = "layer%d.%d" % (li, bi)
            conv(b + ".conv1", cout, cin, 3)
            bn(b + ".bn1", cout)
            conv(b + ".conv2", cout, cout, 3)
            bn(b + ".bn2", cout)
            if cin != cout:
                conv(b + ".downsample.0", cout, cin, 1)
                bn(b + ".downsample.1", cout)
            cin = cout
    return sd


def lower_deepsort_net(sd, in_h=128, in_w=64):
    """the reference's DeepSORT embedding network (reid_models/deepsort_reid.py:14-110, Net(reid=True)) as the same op list: conv 3x3 + BN + ReLU,
    MaxPool(3, 2, 1), four stages of two BasicBlocks (64, 128, 256, 512; the first block of stages 2-4 strides by 2 and projects the shortcut with a
    strided 1x1 + BN), AvgPool2d((8, 4)) -- the whole 8 x 4 map of a 128 x 64 crop --, x / |x|"""
    if (in_h, in_w) != (128, 64):
        raise ValueError("deepsort_reid.Net pools an 8 x 4 map: crops are 128 x 64 (H x W)")
    L = _Lowering(sd, in_h, in_w)
    x, H, W = L.conv(L.in_buf, in_h, in_w, 3, 64, 3, 1, 1, "conv.0", "conv.1", relu=1, kmajor=True)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = L.buf(Ho * Wo * 64)
    L.op(type=MAXPOOL3S2, in_buf=x, out_buf=y, H=H, W=W, C=64, Ho=Ho, Wo=Wo, Co=64)
    x, H, W, cin = y, Ho, Wo, 64
    for li, cout in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for bi in range(2):
            b = "layer%d.%d" % (li, bi)
            s = 2 if (bi == 0 and li > 1) else 1
            y1, H1, W1 = L.conv(x, H, W, cin, cout, 3, s, 1, b + ".conv1", b + ".bn1", relu=1, kmajor=True)
            y2, _, _ = L.conv(y1, H1, W1, cout, cout, 3, 1, 1, b + ".conv2", b + ".bn2", kmajor=True)
            idn = x
            if b + ".downsample.0.weight" in sd:
                idn, _, _ = L.conv(x, H, W, cin, cout, 1, s, 0, b + ".downsample.0", b + ".downsample.1", kmajor=True)
            out = L.buf(H1 * W1 * cout)
            L.op(type=ADD_RELU, in_buf=y2, aux_buf=idn, out_buf=out, H=H1, W=W1, C=cout)
            x, H, W, cin = out, H1, W1, cout
    assert (H, W) == (8, 4)
    v = L.buf(512)
    L.op(type=GAP, in_buf=x, out_buf=v, H=H, W=W, C=512)
    out = L.buf(512)
    L.op(type=L2NORM, in_buf=v, out_buf=out, H=1, W=1, C=512, Co=512)
    return np.array(L.ops, dtype=OP_DTYPE), L.bufs, np.concatenate(L.w)


def lower_deepsort_net_f16(sd, in_h=128, in_w=64):
    """the same network for the MFMA path: fp16 NHWC activations, every convolution one launch of the detector's implicit-GEMM kernels
    (include/y7t.h: Y7T_REID_H_*).  BatchNorm folded; weights fp16 [Cout][round_up(9 Cin, 64)] in (kh, kw, ci) order; the 3-channel crop is
    padded to 16 channels; the strided 1x1 shortcut projections are written as 3x3 / stride 2 / pad 1 filters whose only non-zero tap is the
    centre (the same pixels: 2y - 1 + 1 = 2y), so every launch is a shape the conv kernels are tested on; ReLU is applied by the consumer
    side kernels (pool, add) or an in-place pass"""
    if (in_h, in_w) != (128, 64):
        raise ValueError("deepsort_reid.Net pools an 8 x 4 map: crops are 128 x 64 (H x W)")
    L = _Lowering(sd, in_h, in_w)
    hbuf = lambda halves: L.buf((halves + 1) // 2)            # buffer sizes are in floats per crop

    def hconv(x, H, W, cin, cin_pad, cout, k, s, p, wname, bnname, centre_tap_of_3x3=False):
        Wt = sd[wname + ".weight"].double().numpy()
        scale, bias = L._bn(bnname)
        Wt = Wt * scale[:, None, None, None]
        if wname + ".bias" in sd:
            bias = bias + scale * sd[wname + ".bias"].double().numpy()
        if centre_tap_of_3x3:                                  # 1x1 / stride s / pad 0 == 3x3 / stride s / pad 1 with the centre tap only
            W3 = np.zeros((cout, cin, 3, 3))
            W3[:, :, 1, 1] = Wt[:, :, 0, 0]
            Wt, k, p = W3, 3, 1
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        K = k * k * cin_pad
        Kp = (K + 63) // 64 * 64
        Wp = np.zeros((cout, k, k, cin_pad))
        Wp[..., :cin] = Wt.transpose(0, 2, 3, 1)
        blk = np.zeros((cout, Kp), np.float16)
        blk[:, :K] = Wp.reshape(cout, K).astype(np.float16)
        y = hbuf(Ho * Wo * cout)
        L.op(type=H_CONV, in_buf=x, out_buf=y, H=H, W=W, C=cin_pad, Ho=Ho, Wo=Wo, Co=cout, k=k, s=s, p=p,
             w_off=L.put(blk.reshape(-1).view(np.float32)), b_off=L.put(bias))
        return y, Ho, Wo
    x = hbuf(in_h * in_w * 16)
    L.op(type=H_PACK, in_buf=L.in_buf, out_buf=x, H=in_h, W=in_w, C=3, Co=16)
    x, H, W = hconv(x, in_h, in_w, 3, 16, 64, 3, 1, 1, "conv.0", "conv.1")
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = hbuf(Ho * Wo * 64)
    L.op(type=H_MAXPOOL_RELU, in_buf=x, out_buf=y, H=H, W=W, C=64, Ho=Ho, Wo=Wo, Co=64)
    x, H, W, cin = y, Ho, Wo, 64
    for li, cout in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for bi in range(2):
            b = "layer%d.%d" % (li, bi)
            s = 2 if (bi == 0 and li > 1) else 1
            y1, H1, W1 = hconv(x, H, W, cin, cin, cout, 3, s, 1, b + ".conv1", b + ".bn1")
            L.op(type=H_RELU, in_buf=y1, out_buf=y1, H=H1, W=W1, C=cout)
            y2, _, _ = hconv(y1, H1, W1, cout, cout, cout, 3, 1, 1, b + ".conv2", b + ".bn2")
            idn = x
            if b + ".downsample.0.weight" in sd:
                idn, _, _ = hconv(x, H, W, cin, cin, cout, 1, s, 0, b + ".downsample.0", b + ".downsample.1", centre_tap_of_3x3=True)
            out = hbuf(H1 * W1 * cout)
            L.op(type=H_ADD_RELU, in_buf=y2, aux_buf=idn, out_buf=out, H=H1, W=W1, C=cout)
            x, H, W, cin = out, H1, W1, cout
    assert (H, W) == (8, 4)
    out = L.buf(512)
    L.op(type=H_GAP_L2NORM, in_buf=x, out_buf=out, H=H, W=W, C=512, Co=512)
    return np.array(L.ops, dtype=OP_DTYPE), L.bufs, np.concatenate(L.w)


def _frags(Wm, cout_p, cin_p):
    """(cout, cin) matrix -> A operands of v_mfma_f32_16x16x16_f16, [cout_p/16][cin_p/16][lane][4] fp16: lane holds row lane % 16,
    k = 4 * (lane // 16) + e of its 16 x 16 tile (zero padded)"""
    M = np.zeros((cout_p, cin_p))
    M[:Wm.shape[0], :Wm.shape[1]] = Wm
    lane = np.arange(64)
    out = np.empty((cout_p // 16, cin_p // 16, 64, 4), np.float16)
    for g in range(cout_p // 16):
        for k in range(cin_p // 16):
            for e in range(4):
                out[g, k, :, e] = M[g * 16 + lane % 16, k * 16 + 4 * (lane // 16) + e]
    return out.tobytes()


def _f32(a, n=None):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    if n is not None:
        a = np.concatenate([a, np.zeros(n - a.size)])
    return a.astype(np.float32).tobytes()


def pack_fused(sd, spec):
    """parameters of osnet_x0_25 in the consumption order of csrc/y7t_reid_fused.hip::k_osnet_x025 (BatchNorm folded; 1x1 weights as MFMA fragments;
    the 24-channel bottleneck of stage 3 zero-padded to 32)"""
    assert spec["channels"] == [16, 64, 96, 128] and spec["layers"] == [2, 2, 2] and spec["feature_dim"] == 512
    L = _Lowering(sd, 128, 64)          # only for its BatchNorm folding helper
    out = []

    def conv_bn(name, bnname):
        Wt = sd[name + ".weight"].double().numpy()
        scale, bias = L._bn(bnname)
        return Wt * scale[:, None, None, None], bias

    # conv1 7x7: K-step (kh, half) covers input pixels kw = 4 * half + q (q = lane // 16) x 4 channels (the 4th channel and kw = 7 are zero)
    W1, b1 = conv_bn("conv1.conv", "conv1.bn")
    lane = np.arange(64)
    fr = np.zeros((14, 64, 4), np.float16)
    for kh in range(7):
        for h in range(2):
            kw = 4 * h + lane // 16
            for c in range(3):
                ok = kw < 7
                fr[kh * 2 + h, ok, c] = W1[(lane % 16)[ok], c, kh, kw[ok]]
    out += [fr.tobytes(), _f32(b1)]

    def light(name, mid, midp):
        scale, bias = L._bn(name + ".bn")
        Wd = sd[name + ".conv2.weight"].double().numpy().reshape(mid, 9) * scale[:, None]
        dw = np.zeros((midp // 8, 9, 8))
        for c in range(mid):
            dw[c // 8, :, c % 8] = Wd[c]
        return [_frags(sd[name + ".conv1.weight"].double().numpy().reshape(mid, mid), midp, midp), _f32(dw), _f32(bias, midp)]

    def block(name, cin, cout):
        mid = cout // 4
        midp, R = (16 if mid <= 16 else 32), mid // 16
        Wc1, bc1 = conv_bn(name + ".conv1.conv", name + ".conv1.bn")
        o = [_frags(Wc1.reshape(mid, cin), midp, cin), _f32(bc1, midp)]
        g = name + ".gate"
        w1 = np.zeros((R, midp)); w1[:, :mid] = sd[g + ".fc1.weight"].double().numpy().reshape(R, mid)
        w2 = np.zeros((R, midp)); w2[:, :mid] = sd[g + ".fc2.weight"].double().numpy().reshape(mid, R).T
        o += [_f32(w1), _f32(sd[g + ".fc1.bias"].double().numpy(), 4), _f32(w2), _f32(sd[g + ".fc2.bias"].double().numpy(), midp)]
        Wc3, bc3 = conv_bn(name + ".conv3.conv", name + ".conv3.bn")
        down = cin != cout
        if down:
            Wd, bd = conv_bn(name + ".downsample.conv", name + ".downsample.bn")
            bc3 = bc3 + bd
        o += [_frags(Wc3.reshape(cout, mid), cout, midp), _f32(bc3)]
        if down:
            o.append(_frags(Wd.reshape(cout, cin), cout, cin))
        o += light(name + ".conv2a", mid, midp)
        for tag, n in (("conv2b", 2), ("conv2c", 3), ("conv2d", 4)):
            for j in range(n):
                o += light("%s.%s.%d" % (name, tag, j), mid, midp)
        return o

    def conv1x1(name, c):
        Wc, bc = conv_bn(name + ".conv", name + ".bn")
        return [_frags(Wc.reshape(c, c), c, c), _f32(bc)]
    out += block("conv2.0", 16, 64) + block("conv2.1", 64, 64) + conv1x1("conv2.2.0", 64)
    out += block("conv3.0", 64, 96) + block("conv3.1", 96, 96) + conv1x1("conv3.2.0", 96)
    out += block("conv4.0", 96, 128) + block("conv4.1", 128, 128)
    out += conv1x1("conv5", 128)
    scale, bias = L._bn("fc.1")
    Wf = sd["fc.0.weight"].double().numpy() * scale[:, None]               # (512, 128)
    bf = sd["fc.0.bias"].double().numpy() * scale + bias
    out += [np.ascontiguousarray(Wf.T.reshape(64, 2, 512).transpose(0, 2, 1)).astype(np.float16).tobytes(), _f32(bf)]
    return np.frombuffer(b"".join(out), dtype=np.uint8)


class ReIDExtractor:
    """callable at DeepSORT's reid_model seam.  state_dict: torchreid OSNet names (e.g. torch.load('weights/osnet_x0_25.pth')), or the
    `net_dict` of the reference's own DeepSORT embedding network (arch="deepsort": reid_models/deepsort_reid.py Net, what Extractor loads from
    weights/ckpt.t7), or None for seeded random weights.  size = (W, H) of the network input like Extractor.size (deepsort_reid.py:122)."""

    def __init__(self, state_dict=None, width=0.25, size=(64, 128), max_crops=512, seed=0, fused=None, arch=None, mfma=True):
        """fused: run frame crops through the one-workgroup-per-crop MFMA kernel (fp16 storage, fp32 accumulate).  Default: on for the
        configuration it exists for (OSNet x0_25, 128 x 64 crops); off = the fp32 op list (also what forward_crops always uses).
        arch: "osnet" | "deepsort"; default: read off the state dict's parameter names (OSNet without one).  The deepsort network (1.1 GMAC per
        crop, all of it dense 3x3 convolutions) runs on the detector's MFMA conv kernels with fp16 activations (mfma=True, 4.9 MB of buffers per
        crop) or as the exact fp32 op list (mfma=False, 9.4 MB per crop: keep max_crops near the detections of a frame)."""
        _lib.require_gpu()
        self._L = _lib.load()
        if arch is None:
            arch = "deepsort" if (state_dict is not None and any(k in ("conv.0.weight", "module.conv.0.weight") for k in state_dict)) else "osnet"
        if arch not in ("osnet", "deepsort"):
            raise ValueError("arch %r: osnet or deepsort" % (arch,))
        self.arch = arch
        self.in_w, self.in_h = int(size[0]), int(size[1])
        self.max_crops = int(max_crops)
        if arch == "deepsort":
            self.spec = {"feature_dim": 512}
            if state_dict is None:
                state_dict = deepsort_net_random_state_dict(seed)
        else:
            self.spec = osnet_spec(width)
            if state_dict is None:
                state_dict = random_state_dict(self.spec, seed)
        self.sd = {k.replace("module.", "", 1) if k.startswith("module.") else k: v.detach().float().cpu() for k, v in state_dict.items()}
        self.mfma = bool(mfma) and arch == "deepsort"
        if arch == "deepsort":
            ops, bufs, w = (lower_deepsort_net_f16 if self.mfma else lower_deepsort_net)(self.sd, self.in_h, self.in_w)
        else:
            ops, bufs, w = lower(self.sd, self.spec, self.in_h, self.in_w)
        self.ops, self.feat_dim = ops, self.spec["feature_dim"]
        offs, o = [], 0
        for b in bufs:
            offs.append(o)
            o += (b * self.max_crops + 63) // 64 * 64
        self._offs = np.array(offs, dtype=np.int64)
        self._arena = torch.zeros(o + 64, dtype=torch.float32, device="cuda")
        self._w = torch.from_numpy(w).cuda()
        h = ctypes.c_void_p()
        _lib.check(self._L.y7t_reid_create(ops.ctypes.data_as(ctypes.c_void_p), len(ops), self._offs.ctypes.data_as(ctypes.c_void_p), len(offs),
                                           _lib.ptr(self._arena), self._arena.numel() * 4, _lib.ptr(self._w), self.max_crops, self.in_h, self.in_w,
                                           self.feat_dim, ctypes.byref(h)))
        self._h = h
        can_fuse = arch == "osnet" and width == 0.25 and (self.in_w, self.in_h) == (64, 128)
        if fused and not can_fuse:
            raise _lib.Y7TError("the fused ReID kernel is OSNet x0_25 on 128 x 64 crops (got width %s, size %s)" % (width, size))
        self.fused = can_fuse if fused is None else bool(fused)
        if self.fused:
            blob = pack_fused(self.sd, self.spec)
            assert blob.size == self._L.y7t_reid_fused_blob_size(), (blob.size, self._L.y7t_reid_fused_blob_size())
            self._blob = torch.from_numpy(blob.copy()).cuda()
            _lib.check(self._L.y7t_reid_set_fused(self._h, _lib.ptr(self._blob), self._blob.numel()))

    @classmethod
    def from_checkpoint(cls, path, **kw):
        """torchreid-style OSNet checkpoint (what the reference's load_pretrained_weights reads, reid_models/load_model_tools.py): a state
        dict, or {'state_dict': ...}, with or without the DataParallel 'module.' prefix; classifier.* is ignored.  The width is read off
        conv1's channel count."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        if isinstance(ck, dict) and "net_dict" in ck:       # Extractor.__init__ (deepsort_reid.py:115-117): weights/ckpt.t7 of the original DeepSORT
            sd = {k: v for k, v in ck["net_dict"].items() if not k.startswith("classifier")}
            kw.setdefault("max_crops", 128)
            return cls(sd, arch="deepsort", **kw)
        sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck.state_dict()
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items() if not k.startswith(("classifier", "module.classifier"))}
        if "conv1.conv.weight" not in sd:
            raise _lib.Y7TError("%s is neither an OSNet checkpoint (conv1.conv.weight) nor a DeepSORT one ({'net_dict': ...})" % path)
        width = {64: 1.0, 48: 0.75, 32: 0.5, 16: 0.25}[int(sd["conv1.conv.weight"].shape[0])]
        return cls(sd, width=width, **kw)

    def features_for_boxes(self, frame, tlbrs):
        """frame: (H, W, 3) uint8 BGR (host or device); tlbrs: (N, 4) -> (N, feat_dim) float32 DEVICE tensor"""
        if not isinstance(frame, torch.Tensor):
            frame = torch.from_numpy(np.ascontiguousarray(frame))
        frame = frame.to(device="cuda", dtype=torch.uint8).contiguous()
        boxes = torch.as_tensor(np.ascontiguousarray(np.asarray(tlbrs, dtype=np.float32).reshape(-1, 4))).cuda()
        n = boxes.shape[0]
        out = torch.empty((n, self.feat_dim), dtype=torch.float32, device="cuda")
        for i in range(0, n, self.max_crops):       # more boxes than the arena holds: max_crops at a time (stream order keeps the buffers safe)
            m = min(self.max_crops, n - i)
            _lib.check(self._L.y7t_reid_forward(self._h, _lib.ptr(frame), frame.shape[0], frame.shape[1], _lib.ptr(boxes[i:]), m, None, _lib.ptr(out[i:]),
                                                _lib.stream_ptr()))
        self._keep = (frame, boxes)
        return out

    def features_for_frames(self, frames, tlbrs, frame_idx):
        """crops from a batch of frames in one pass: frames (B, H, W, 3) uint8 device tensor, tlbrs (N, 4) float32 and frame_idx (N) int32
        device tensors (or arrays) -> (N, feat_dim) float32 device tensor"""
        frames = frames.to(device="cuda", dtype=torch.uint8).contiguous()
        boxes = torch.as_tensor(tlbrs, dtype=torch.float32).reshape(-1, 4).cuda().contiguous()
        idx = torch.as_tensor(frame_idx, dtype=torch.int32).reshape(-1).cuda().contiguous()
        n = boxes.shape[0]
        if n > self.max_crops or idx.shape[0] != n:
            raise _lib.Y7TError("%d crops (max_crops=%d), %d frame indices" % (n, self.max_crops, idx.shape[0]))
        out = torch.empty((n, self.feat_dim), dtype=torch.float32, device="cuda")
        _lib.check(self._L.y7t_reid_forward_batch(self._h, _lib.ptr(frames), frames.shape[0], frames.shape[1], frames.shape[2], _lib.ptr(boxes), _lib.ptr(idx), n,
                                                  _lib.ptr(out), _lib.stream_ptr()))
        self._keep = (frames, boxes, idx)
        return out

    def forward_crops(self, crops_nhwc):
        """(N, in_h, in_w, 3) float32 crops already resized + normalised (tests) -> (N, feat_dim) device tensor"""
        x = crops_nhwc.to(device="cuda", dtype=torch.float32).contiguous()
        n = x.shape[0]
        out = torch.empty((n, self.feat_dim), dtype=torch.float32, device="cuda")
        _lib.check(self._L.y7t_reid_forward(self._h, None, 0, 0, None, n, _lib.ptr(x), _lib.ptr(out), _lib.stream_ptr()))
        self._keep = x
        return out

    def __call__(self, im_crops):
        """Extractor.__call__ (deepsort_reid.py:148-153) for host crops: list of (h, w, 3) uint8 BGR arrays -> (N, feat_dim) numpy.
        The crops are packed side by side into one frame so that the device crop kernel does the resize."""
        if not im_crops:
            return np.zeros((0, self.feat_dim), np.float32)
        hmax, wsum = max(c.shape[0] for c in im_crops), sum(c.shape[1] for c in im_crops)
        canvas = np.zeros((hmax, wsum, 3), np.uint8)
        boxes, x = [], 0
        for c in im_crops:
            canvas[:c.shape[0], x:x + c.shape[1]] = c
            boxes.append((x, 0, x + c.shape[1], c.shape[0]))
            x += c.shape[1]
        return self.features_for_boxes(canvas, np.asarray(boxes, np.float32)).cpu().numpy()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.y7t_reid_destroy(self._h)
        except Exception:
            pass
