"""Graph parse + lowering.

`parse(spec)` restates the subset of /root/reference/models/yolo.py:443-520 (parse_model) that the yolov7-w6 /
yolov7-tiny graphs use -- Conv, ReOrg, Concat, MP, SP, SPPCSPC, nn.Upsample, Detect / IDetect / IAuxDetect -- and
expands SPPCSPC (models/common.py:262-280) into its 7 convs + 3 pools.  `lower(nodes, H, W)` turns the node list into
the static launch list of include/y7t.h (`y7t_op`) over an NHWC fp16 arena:
  * every tensor lives in exactly one buffer; a tensor consumed by a Concat lives INSIDE the concat's buffer at its
    channel offset, so Concat (models/common.py:56-62) costs nothing (17 concats in w6);
  * ReOrg is fused into the input layout kernel; branches that do not reach the Detect head (the aux head of
    training checkpoints, models/yolo.py:141-153) are dropped.
"""
import math

import os

import numpy as np

from .. import _lib


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


class Node:
    """one tensor-producing op after expansion. kind: input|reorg|conv|concat|up|pool|detect"""
    __slots__ = ("idx", "kind", "src", "c", "k", "s", "p", "act", "wkey", "layer", "extra", "h", "w", "home", "coff", "ld", "virt_up", "virtual")

    def __init__(self, kind, src, c, k=1, s=1, p=0, act=0, wkey=None, layer=-1, extra=None):
        self.kind, self.src, self.c, self.k, self.s, self.p, self.act, self.wkey, self.layer, self.extra = kind, list(src), c, k, s, p, act, wkey, layer, extra
        self.idx = -1
        self.h = self.w = 0
        self.home, self.coff, self.ld = None, 0, 0
        self.virt_up, self.virtual = None, False


def _act_code(a):
    if a is None or a is True:
        return 1  # SiLU (Conv default act=True)
    s = str(a)
    if "LeakyReLU" in s:
        return 2
    if s in ("False", "nn.Identity()", "Identity()"):
        return 0
    if "SiLU" in s:
        return 1
    raise NotImplementedError("activation %r" % (a,))


def parse(spec, ch=3):
    """-> (nodes, layer_out) : expanded node list and, per reference layer index, the node that is its output"""
    nc, anchors, gw = spec["nc"], spec["anchors"], spec.get("width_multiple", 1.0)
    na = len(anchors[0]) // 2
    no = na * (nc + 5)
    nodes = [Node("input", [], ch)]
    layer_out, chs = [], []

    def add(n):
        n.idx = len(nodes)
        nodes.append(n)
        return n.idx

    for i, (f, n, m, args) in enumerate(spec["layers"]):
        if n != 1:
            raise NotImplementedError("layer %d: number=%d" % (i, n))
        fl = [f] if isinstance(f, int) else list(f)
        src = [(0 if i == 0 else layer_out[i - 1]) if x == -1 else layer_out[x if x >= 0 else i + x] for x in fl]
        cin = [(ch if i == 0 else chs[i - 1]) if x == -1 else chs[x if x >= 0 else i + x] for x in fl]
        m = str(m)
        if m == "Conv":
            c2 = args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            k = args[1] if len(args) > 1 else 1
            s = args[2] if len(args) > 2 else 1
            p = args[3] if len(args) > 3 and args[3] not in (None, "None") else k // 2   # yaml spells it 'None'
            if len(args) > 4 and args[4] not in (1, None):
                raise NotImplementedError("grouped conv")
            act = _act_code(args[5] if len(args) > 5 else True)
            out = add(Node("conv", src, c2, k, s, p, act, "model.%d" % i, i))
        elif m == "ReOrg":
            out, c2 = add(Node("reorg", src, cin[0] * 4, layer=i)), cin[0] * 4
        elif m == "Concat":
            out, c2 = add(Node("concat", src, sum(cin), layer=i)), sum(cin)
        elif m == "MP":
            k = args[0] if args else 2
            out, c2 = add(Node("pool", src, cin[0], k, k, 0, layer=i)), cin[0]
        elif m == "SP":
            k = args[0] if args else 3
            s = args[1] if len(args) > 1 else 1
            out, c2 = add(Node("pool", src, cin[0], k, s, k // 2, layer=i)), cin[0]
        elif m in ("nn.Upsample", "Upsample"):
            if args[0] not in (None, "None") or args[1] != 2 or args[2] != "nearest":
                raise NotImplementedError("Upsample%r" % (args,))
            out, c2 = add(Node("up", src, cin[0], layer=i)), cin[0]
        elif m == "SPPCSPC":
            c2 = make_divisible(args[0] * gw, 8)
            c_ = int(2 * c2 * 0.5)
            key = "model.%d." % i
            cv1 = add(Node("conv", src, c_, 1, 1, 0, 1, key + "cv1", i))
            cv3 = add(Node("conv", [cv1], c_, 3, 1, 1, 1, key + "cv3", i))
            cv4 = add(Node("conv", [cv3], c_, 1, 1, 0, 1, key + "cv4", i))
            # max-pool 5/9/13 (stride 1, -inf padding) == 5, 5o5, 5o5o5
            p5 = add(Node("pool", [cv4], c_, 5, 1, 2, layer=i))
            p9 = add(Node("pool", [p5], c_, 5, 1, 2, layer=i))
            p13 = add(Node("pool", [p9], c_, 5, 1, 2, layer=i))
            cat1 = add(Node("concat", [cv4, p5, p9, p13], 4 * c_, layer=i))
            cv5 = add(Node("conv", [cat1], c_, 1, 1, 0, 1, key + "cv5", i))
            cv6 = add(Node("conv", [cv5], c_, 3, 1, 1, 1, key + "cv6", i))
            cv2 = add(Node("conv", src, c_, 1, 1, 0, 1, key + "cv2", i))
            cat2 = add(Node("concat", [cv6, cv2], 2 * c_, layer=i))
            out = add(Node("conv", [cat2], c2, 1, 1, 0, 1, key + "cv7", i))
        elif m in ("Detect", "IDetect", "IAuxDetect"):
            nl = len(anchors)
            out = add(Node("detect", src[:nl], no, layer=i, extra={"kind": m, "nl": nl, "na": na, "no": nc + 5, "cin": cin[:nl]}))
            c2 = no
        else:
            raise NotImplementedError("module %s (layer %d) is outside the yolov7-w6 / yolov7-tiny hot path" % (m, i))
        layer_out.append(out)
        chs.append(c2)
    return nodes, layer_out


OP_DTYPE = np.dtype([("type", "<i4"), ("in_buf", "<i4"), ("in_ld", "<i4"), ("in_coff", "<i4"), ("H", "<i4"), ("W", "<i4"), ("Cin", "<i4"),
                     ("out_buf", "<i4"), ("out_ld", "<i4"), ("out_coff", "<i4"), ("out_f32", "<i4"), ("Ho", "<i4"), ("Wo", "<i4"),
                     ("Cout", "<i4"), ("Cout_pad", "<i4"), ("KH", "<i4"), ("KW", "<i4"), ("stride", "<i4"), ("pad", "<i4"), ("K", "<i4"),
                     ("K_pad", "<i4"), ("act", "<i4"), ("korder", "<i4"), ("detect_level", "<i4"), ("w_off", "<i8"), ("bias_off", "<i8"),
                     ("up_buf", "<i4"), ("up_ld", "<i4"), ("up_coff", "<i4"), ("up_c0", "<i4"), ("up_C", "<i4"), ("pad0", "<i4")], align=False)
assert OP_DTYPE.itemsize == 136   # == sizeof(y7t_op) in include/y7t.h


def _rup(x, m):
    return (x + m - 1) // m * m


class Plan:
    """result of `lower`: ops (structured array), buffer table, weight layout, head description"""


def patch_panel_rows(H, W, cout, B):
    """rows of a weight panel (= output channels of a workgroup) of the LDS-patch kernel: 128 when Cout allows -- unless that leaves fewer than 512 workgroups of
    256 pixels for the whole batch: then 64 (korder 9), twice the workgroups.  Measured at 32 frames (round 4, profiles/r04_smallmap_patch.txt): the 20x20 layers with
    256 output channels 71 -> 51 us and 39 -> 28 us against the generic kernel, 61 / 36 us with 128-row panels.  Y7T_CONV_PATCH_PANEL64_BELOW=0 switches the rule off."""
    cout_pad = -(-cout // 64) * 64
    if cout_pad % 128:
        return 64
    below = int(_lib.switch("Y7T_CONV_PATCH_PANEL64_BELOW", "512"))      # (round 6, 80 frames: the 20 x 20 512 -> 512 layers at 500 tiles 167 -> 159 us on 64-row panels; round 5 at 40 frames: 250 tiles, 104 -> 95 us)
    return 64 if B * H * W // 256 * (cout_pad // 128) < below else 128


# fewest workgroups (of 256 pixels x one weight panel) the LDS-patch kernel is given.  Round 3: 256.  Round 4 measured the 20x20 layers at 32 frames (200 ... 488 workgroups:
# 91 -> 70, 39 -> 28 us, profiles/r04_smallmap_patch.txt) and the batch-1 list (profiles/r04_latency_lowering.txt): against the generic kernel + split-K the patch kernel
# wins from 100 workgroups (80^2 256 -> 256: 34.8 -> 26.3 us), and from 50 when K is short (Cin <= 128: 80^2 128 -> 128 20.8 -> 15.9, 160^2 64 -> 64 20.4 -> 11.8);
# with 50 workgroups and Cin = 256 it loses (23.5 -> 26.0 us), as does the 40-wide strip with 84 (35.8 -> 38.1)
PATCH_MIN_PIX = 100 * 256
PATCH_MIN_PIX_SHALLOW = 50 * 256


def patch_eligible(H, W, cin, cout, k, s, p, out_ld, out_coff, out_f32, B=1 << 20):
    """mirror of y7t_conv_patch_try (csrc/y7t_conv_patch.hip): 3x3 / stride 1 / pad 1, Cin % 64 == 0, 16-byte aligned fp16 output,
    a 16x16 / 32x8 / strip tiling that computes at least 80 % useful pixels, and at least 100 (Cin <= 128: 50) workgroups of 256 pixels x 128 (64)
    channels at batch B (below that -- batch-1 latency mode -- the generic kernel with split-K fills the chip better)"""
    cout_pad = -(-cout // 64) * 64
    bn = patch_panel_rows(H, W, cout, B)
    if B * H * W * (cout_pad // bn) < int(_lib.switch("Y7T_CONV_PATCH_MIN_PIX", str(PATCH_MIN_PIX_SHALLOW if cin <= 128 else PATCH_MIN_PIX))):      # (the switch: ONE threshold for the A/B)
        return False
    if _lib.switch("Y7T_CONV_PATCH", "1") == "0" or _lib.switch("Y7T_CONV_VARIANT", "0") != "0":
        return False
    if not (k == 3 and s == 1 and p == 1 and cin % 64 == 0 and not out_f32 and cout % 8 == 0 and out_ld % 8 == 0 and out_coff % 8 == 0):
        return False
    eff = lambda tw, th: (W * H) / float(-(-W // tw) * tw * -(-H // th) * th)
    eflat = (W * H) / float((W + 2) * (H + 2)) if W in (20, 40) else 0.0      # strip tiling of narrow maps
    return max(eff(16, 16), eff(32, 8), eflat) >= 0.8


def patch_s2_eligible(cin, cout, k, s, p, out_ld, out_coff, out_f32, M_out=1 << 30):
    """mirror of y7t_conv_patch_s2_launch (csrc/y7t_conv_patch_s2.hip): the 3x3 / stride-2 down-sampling layers on the parity-split LDS-patch kernel.
    Measured per layer at 32 frames (round 3, profiles/r03_conv_variants.txt): with 256-channel panels and Cin >= 128 it beats the generic kernel on the
    320^2 / 160^2 / 80^2 inputs (672 -> 587, 598 -> 491, 415 -> 395, 188 -> 164 us); with 128-channel panels (Cout = 128, 384) and on the small maps
    (fewer than ~50 000 output pixels per launch: 40^2 inputs, batch-1 latency mode) it loses -- those keep the generic kernel.
    Y7T_CONV_PATCH_S2=0 switches it off; =1 takes every layer the kernel can run (the experiment's rule; Y7T_CONV_PATCH_S2_MIN_COUT, _BN as before)."""
    mode = _lib.switch("Y7T_CONV_PATCH_S2", "auto")
    if mode == "0" or _lib.switch("Y7T_CONV_VARIANT", "0") != "0":
        return False
    cout_pad = -(-cout // 64) * 64
    can = (k == 3 and s == 2 and p == 1 and cin % 64 == 0 and cout_pad % 128 == 0 and not out_f32 and cout % 8 == 0 and out_ld % 8 == 0
           and out_coff % 8 == 0)
    if mode == "1":
        return can and cout_pad >= int(_lib.switch("Y7T_CONV_PATCH_S2_MIN_COUT", "128"))
    return can and cout_pad % 256 == 0 and cin >= 128 and M_out >= 50000 and _lib.switch("Y7T_CONV_PATCH_S2_BN", "") != "128"


def ws_eligible(H, W, cin, cout, k, s, p, out_ld, out_coff, out_f32, in_ld, in_coff, B=1 << 20):
    """mirror of y7t_conv_ws_launch (csrc/y7t_conv_ws.hip): the 64 -> 64 3x3 / stride 1 layers with the whole filter bank resident in registers and a
    persistent workgroup per compute unit.  Measured at 32 frames (round 3): 320^2 358 -> 312 us, 160^2 91 -> 78 us per layer against the multi-tile patch
    kernel (Y7T_CONV_WS=0 switches back); needs enough 16 x 16 tiles to give every compute unit a few (below that -- batch-1 latency mode -- the launch
    keeps its current kernel)."""
    if _lib.switch("Y7T_CONV_WS", "1") == "0" or _lib.switch("Y7T_CONV_VARIANT", "0") != "0":
        return False
    tiles = B * -(-H // 16) * -(-W // 16)
    return (k == 3 and s == 1 and p == 1 and cin == 64 and cout == 64 and H % 16 == 0 and W % 16 == 0 and not out_f32 and out_ld % 8 == 0 and out_coff % 8 == 0 and in_ld % 8 == 0
            and in_coff % 8 == 0 and tiles >= int(_lib.switch("Y7T_CONV_WS_MIN_TILES", "1024")))


def ws_s2_eligible(H, W, cin, cout, k, s, p, out_ld, out_coff, out_f32, in_ld, in_coff, B=1 << 20):
    """mirror of y7t_conv_ws_s2_launch (csrc/y7t_conv_ws_s2.hip): the 64 -> 128 3x3 / stride 2 layer (the first down-sampling conv of yolov7-w6) with its filter bank in
    registers and a persistent workgroup per compute unit; needs enough 2 x 32 output tiles to give every compute unit a few.  Y7T_CONV_WS_S2=0 switches it off."""
    if _lib.switch("Y7T_CONV_WS_S2", "1") == "0" or _lib.switch("Y7T_CONV_VARIANT", "0") != "0":
        return False
    Ho, Wo = H // 2, W // 2
    return (k == 3 and s == 2 and p == 1 and cin == 64 and cout == 128 and H % 2 == 0 and W % 2 == 0 and Ho % 2 == 0 and Wo % 32 == 0 and not out_f32
            and out_ld % 8 == 0 and out_coff % 8 == 0 and in_ld % 8 == 0 and in_coff % 8 == 0 and B * (Ho // 2) * (Wo // 32) >= int(_lib.switch("Y7T_CONV_WS_MIN_TILES", "1024")))


def p8_eligible(H, W, cin, cout, k, s, out_ld, out_coff, out_f32, in_ld, in_coff, B=1 << 20, up=None, detect=False):
    """mirror of y7t_conv_p8_launch (csrc/y7t_conv_p8.hip): 1x1 layers with Cout % 256 == 0 on the 256 x 256 x 64 ping-pong pipeline -- WHERE IT MEASURED FASTER than
    igemm<128,128,32,2> on two boxes (profiles/r04_p8_measurements.txt): deep reductions (Cin >= 1024: 16+ K-tiles amortise the tile's ~8 us of prologue / epilogue /
    drain) and Cin >= 512 on grids of >= 3000 tiles (many rounds: little quantisation loss).  The shallower layers tie or lose and stay on the 128-pixel tiles.
    up = (up_c0, up_C) of an upsample-on-read layer.  Y7T_CONV_P8=0 switches it off, Y7T_CONV_P8=all takes every eligible shape (A/B inside one session)."""
    mode = _lib.switch("Y7T_CONV_P8", "1")
    if mode == "0" or _lib.switch("Y7T_CONV_VARIANT", "0") != "0" or detect:
        return False
    tiles = -(-(B * H * W) // 256) * (cout // 256 if cout % 256 == 0 else 0)
    ok = (k == 1 and s == 1 and cin % 64 == 0 and cout % 256 == 0 and not out_f32 and out_ld % 8 == 0 and out_coff % 8 == 0 and in_ld % 8 == 0 and in_coff % 8 == 0
          and tiles >= int(_lib.switch("Y7T_CONV_P8_MIN_TILES", "230")))
    if mode != "all" and _lib.switch("Y7T_CONV_P8_MIN_TILES", None) is None:      # (through the same gate as the threshold: an experiment variable in the environment does not change the PRODUCT's lowering)
        fill = tiles / float(max(1, -(-tiles // 256)) * 256)      # how full the last round of one-tile-per-CU workgroups is on 256 CUs
        # >= 1500 tiles (six rounds of the chip), or Cin >= 512 on a grid that fills its rounds to >= 90 % (the 20 x 20 512-output-channel layers at 80 frames: 250 tiles):
        # round 6 at 80 frames, every such launch 2-27 % ahead of igemm<128,128,32,2>; 40 x 40 384 -> 256 (500 tiles, six K-tiles) loses 8 % and stays (profiles/r06_batch_80.txt)
        ok = ok and (cin >= 1024 or (cin >= 512 and tiles >= 3000) or tiles >= 1500 or (cin >= 512 and fill >= 0.9))
    if up is not None:
        ok = ok and up[0] % 64 == 0 and up[1] % 64 == 0 and H % 2 == 0 and W % 2 == 0
    return bool(ok)


WS128_DEFAULT = "1"      # round 5, on the tile counter (profiles/r05_tile_counter.txt): the list 14.77 / 14.80 -> 14.70 / 14.64 ms in the pipeline, 14.22 -> 14.05 ms alone


def ws128_eligible(H, W, cin, cout, k, s, p, out_ld, out_coff, out_f32, in_ld, in_coff, B=1 << 20):
    """mirror of y7t_conv_ws128_launch (csrc/y7t_conv_ws128.hip): the 128 -> 128 k 3x3 / stride 1 layers with a 128-channel output tile's filter bank in the
    registers of a persistent workgroup, tiles taken from the op's tile counter.  Round 4 measured its statically partitioned form: 5-19 % faster alone, 0.2 ms slower
    in the pipeline (profiles/r04_ws128_measurement.txt); round 5 on the tile counter: faster in the pipeline too (one session, launch list 14.77 / 14.80 ms without,
    14.70 / 14.64 ms with it; the static form again +0.34 ms: profiles/r05_tile_counter.txt).  Y7T_CONV_WS128=0 switches back to the LDS-patch kernels."""
    if _lib.switch("Y7T_CONV_WS128", WS128_DEFAULT) != "1" or _lib.switch("Y7T_CONV_VARIANT", "0") != "0":
        return False
    tiles = B * (H // 4) * (W // 16)
    return (k == 3 and s == 1 and p == 1 and cin == 128 and cout % 128 == 0 and cout <= 512 and H % 4 == 0 and W % 16 == 0 and not out_f32 and out_ld % 8 == 0
            and out_coff % 8 == 0 and in_ld % 8 == 0 and in_coff % 8 == 0 and tiles >= int(_lib.switch("Y7T_CONV_WS_MIN_TILES", "1024")))


WS128_S2_DEFAULT = "1"      # round 5, sessions r5h / r5i: 786 -> 644 us and 190 -> 154 us per 40 frames alone, the list 18.22 / 18.25 -> 18.12 / 18.10 ms in the pipeline


def ws128_s2_eligible(H, W, cin, cout, k, s, p, out_ld, out_coff, out_f32, in_ld, in_coff, B=1 << 20):
    """mirror of y7t_conv_ws128_launch's stride-2 form (csrc/y7t_conv_ws128.hip, S2): the 3x3 / stride 2 layers with 128 input channels (w6: 128 -> 256 at 320 x 320 and at
    160 x 160) with the filter bank of a 128-channel output tile in registers, 2 x 16 output tiles from the op's tile counter.  Same weight order as the stride-1 kernel
    (korder 6).  Follows Y7T_CONV_WS128 (=0 switches both forms off); Y7T_CONV_WS128_S2=0 alone is an experiment switch (measuring build)."""
    if _lib.switch("Y7T_CONV_WS128_S2", WS128_S2_DEFAULT) != "1" or _lib.switch("Y7T_CONV_WS128", WS128_DEFAULT) != "1" or _lib.switch("Y7T_CONV_VARIANT", "0") != "0":
        return False
    Ho, Wo = H // 2, W // 2
    return (k == 3 and s == 2 and p == 1 and cin == 128 and cout % 128 == 0 and cout <= 512 and H % 2 == 0 and W % 2 == 0 and Ho % 2 == 0 and Wo % 16 == 0 and not out_f32
            and out_ld % 8 == 0 and out_coff % 8 == 0 and in_ld % 8 == 0 and in_coff % 8 == 0 and B * (Ho // 2) * (Wo // 16) >= int(_lib.switch("Y7T_CONV_WS_MIN_TILES", "1024")))


def lower(nodes, H, W, max_batch=1):
    det = next(n for n in nodes if n.kind == "detect")
    # ---- liveness: only what reaches the (main) Detect inputs ----
    live = set()
    stack = list(det.src)
    while stack:
        i = stack.pop()
        if i in live:
            continue
        live.add(i)
        stack.extend(nodes[i].src)
    # ---- shapes ----
    for n in nodes:
        if n.kind == "input":
            n.h, n.w = H, W
            continue
        if n.kind == "detect":
            continue
        s0 = nodes[n.src[0]]
        if n.kind == "reorg":
            n.h, n.w = s0.h // 2, s0.w // 2
        elif n.kind in ("conv", "pool"):
            n.h, n.w = (s0.h + 2 * n.p - n.k) // n.s + 1, (s0.w + 2 * n.p - n.k) // n.s + 1
        elif n.kind == "up":
            n.h, n.w = s0.h * 2, s0.w * 2
        elif n.kind == "concat":
            n.h, n.w = s0.h, s0.w
            for j in n.src:
                assert (nodes[j].h, nodes[j].w) == (n.h, n.w), "concat of mismatched maps"
    # ---- homes: a tensor consumed by a concat lives inside that concat's buffer ----
    bufs = []  # (elems_per_image, itemsize)

    def new_buf(n, ld, itemsize=2):
        bufs.append((n.h * n.w * ld, itemsize))
        return len(bufs) - 1
    extra_copies = []  # (src_node, concat_node, coff) for tensors that sit in more than one concat
    for n in nodes:
        if n.idx in live and n.kind == "concat":
            if any(n.idx in nodes[j].src for j in live if nodes[j].kind == "concat"):
                raise NotImplementedError("nested Concat")
            n.home, n.coff, n.ld = new_buf(n, n.c), 0, n.c
            off = 0
            for j in n.src:
                t = nodes[j]
                if t.home is None and t.kind not in ("input", "reorg", "concat"):
                    t.home, t.coff, t.ld = n.home, off, n.c
                else:
                    extra_copies.append((j, n.idx, off))
                off += t.c
    pending_src = {j for j, _, _ in extra_copies}
    first = nodes[1] if len(nodes) > 1 and nodes[1].kind == "reorg" else None
    in_ld = 16 if first is not None else 8
    nodes[0].home, nodes[0].coff, nodes[0].ld = None, 0, 0
    bufs_input = None
    for n in nodes:
        if n.idx not in live and n.kind != "input":
            continue
        if n.kind in ("input", "reorg"):
            continue
        if n.home is None and n.kind != "detect":
            n.home, n.coff, n.ld = new_buf(n, n.c), 0, n.c
    # buffer 0 must be the input layout buffer: build the final table with it first
    img_node = first if first is not None else nodes[0]
    img_h, img_w = (H // 2, W // 2) if first is not None else (H, W)
    table = [(img_h * img_w * in_ld, 2)] + bufs
    shift = 1
    for n in nodes:
        if n.home is not None:
            n.home += shift
    img_node.home, img_node.coff, img_node.ld = 0, 0, in_ld
    if first is not None:
        nodes[0].home = None
    # ---- upsample-on-read (cfg/deploy/yolov7-w6.yaml:75,89,103): an nn.Upsample whose only consumer is a Concat that only 1x1 / stride-1
    # convs read is never materialised -- those convs fetch its channel range from the half-resolution tensor at (y >> 1, x >> 1) ----
    if _lib.switch("Y7T_UPSAMPLE_ON_READ", "1") != "0":
        users = {}
        for m in nodes:
            if m.idx in live or m.kind == "detect":
                for j in m.src:
                    users.setdefault(j, []).append(m)
        for n in nodes:
            if n.idx not in live or n.kind != "up":
                continue
            cons = users.get(n.idx, [])
            if len(cons) != 1 or cons[0].kind != "concat" or cons[0].virt_up is not None:
                continue
            c = cons[0]
            readers = users.get(c.idx, [])
            off = sum(nodes[j].c for j in c.src[:c.src.index(n.idx)])
            t = nodes[n.src[0]]
            ok = readers and all(r.kind == "conv" and r.k == 1 and r.s == 1 and r.p == 0 for r in readers) and c.c % 64 == 0 and \
                n.c % 32 == 0 and off % 32 == 0 and n.home == c.home and t.home is not None and t.ld % 8 == 0 and t.coff % 8 == 0 and \
                n.h % 2 == 0 and n.w % 2 == 0 and n.idx not in pending_src
            if ok:
                c.virt_up, n.virtual = (n, off), True
    # ---- ops ----
    ops, wlayout = [], []   # wlayout: (wkey, cin_real, cin_pad, cout, cout_pad, k, K, K_pad, w_off, b_off, kind)
    w_off = b_off = 0
    heads = []

    def emit_conv(n, src, out_buf, out_ld, out_coff, out_f32, cout, act, wkey, kind="conv", level=-1, fuse_twin=None):
        """fuse_twin = (a, b): `n` is the 64 -> 128 stride-2 layer and its only consumers are the twin 1x1 convs a, b -- ONE op (korder 11, csrc/y7t_conv_ws_s2.hip) whose
        output is the twins' concat slice; the tensor between them is never written.  Two wlayout entries (the 3x3 bank, then the 1x1 bank: adjacent in the blobs)."""
        nonlocal w_off, b_off
        cin_real = src.c
        cin = _rup(cin_real, 8) if src.kind not in ("input", "reorg") else in_ld
        if src.kind not in ("input", "reorg") and cin != cin_real:
            raise NotImplementedError("conv input channels %d not a multiple of 8" % cin_real)
        K = n.k * n.k * cin
        K_pad = _rup(K, 64)
        cout_pad = _rup(cout, 64)
        op = np.zeros((), OP_DTYPE)
        op["type"], op["in_buf"], op["in_ld"], op["in_coff"] = 0, src.home, src.ld, src.coff
        op["H"], op["W"], op["Cin"] = src.h, src.w, cin
        op["out_buf"], op["out_ld"], op["out_coff"], op["out_f32"] = out_buf, out_ld, out_coff, out_f32
        op["Ho"], op["Wo"], op["Cout"], op["Cout_pad"] = n.h, n.w, cout, cout_pad
        op["KH"], op["KW"], op["stride"], op["pad"], op["K"], op["K_pad"], op["act"] = n.k, n.k, n.s, n.p, K, K_pad, act
        korder = int(n.k == 3 and cin % 64 == 0)     # (kh, 64-channel chunk, kw) K order: consecutive K-steps reuse input lines
        if ws_eligible(src.h, src.w, cin, cout, n.k, n.s, n.p, out_ld, out_coff, out_f32, src.ld, src.coff, max_batch):
            korder = 5                               # weights-stationary kernel: the filter bank as MFMA A-fragments (weights.pack_ws)
        elif ws_s2_eligible(src.h, src.w, cin, cout, n.k, n.s, n.p, out_ld, out_coff, out_f32, src.ld, src.coff, max_batch):
            korder = 8                               # ... and its stride-2 sibling for the 64 -> 128 down-sampling layer (weights.pack_ws_s2)
        elif ws128_eligible(src.h, src.w, cin, cout, n.k, n.s, n.p, out_ld, out_coff, out_f32, src.ld, src.coff, max_batch):
            korder = 6                               # ... and its 128-channel sibling (weights.pack_ws128)
        elif ws128_s2_eligible(src.h, src.w, cin, cout, n.k, n.s, n.p, out_ld, out_coff, out_f32, src.ld, src.coff, max_batch):
            korder = 6                               # ... and the stride-2 form of that kernel (same packing)
        elif patch_eligible(src.h, src.w, cin, cout, n.k, n.s, n.p, out_ld, out_coff, out_f32, max_batch):
            korder = 2                               # LDS-patch kernel: weights in its panel order (weights.panel_pack)
            if cout_pad % 128 == 0 and patch_panel_rows(src.h, src.w, cout, max_batch) == 64:
                korder = 9                           # ... with 64-row panels
        elif patch_s2_eligible(cin, cout, n.k, n.s, n.p, out_ld, out_coff, out_f32, max_batch * n.h * n.w):
            korder = 4                               # stride-2 LDS-patch kernel, weights in its panel order (weights.panel_pack_s2)
        elif p8_eligible(src.h, src.w, cin, cout, n.k, n.s, out_ld, out_coff, out_f32, src.ld, src.coff, max_batch,
                         up=(src.virt_up[1], src.virt_up[0].c) if getattr(src, "virt_up", None) is not None else None, detect=level >= 0):
            korder = 7                               # 1x1, Cout % 256 == 0, a tile per CU: 256 x 64 panels of the ping-pong pipeline (weights.panel_pack_p8)
        elif n.k == 1 and cin % 32 == 0 and _lib.switch("Y7T_CONV_VARIANT", "0") == "0" and _lib.switch("Y7T_CONV_WPANEL", "1") != "0":
            korder = 3                               # 1x1: contiguous per-K-step weight panels (weights.panel_pack_linear)
            if (cout_pad % 128 == 0 and level < 0 and getattr(src, "virt_up", None) is None and
                    -(-max_batch * n.h * n.w // 128) * (cout_pad // 128) < int(_lib.switch("Y7T_CONV_1X1_PANEL64_BELOW", "500"))):
                korder = 10                          # ... as 64-row panels where 128-row tiles number fewer than 500 (round 4, profiles/r04_smallmap_patch.txt: the 20x20
                                                     # layers with <= 512 output channels 34 -> 30, 61 -> 53, 20 -> 18 us; at 800 tiles and above 64 rows lose)
        if fuse_twin is not None:
            if not (korder == 8 and cout == 128 and K_pad == K):
                return False                         # this layer does not get the stride-2 weights-stationary kernel after all (a switch that only this path reads):
                                                     # nothing has been emitted yet -- the caller emits the layer and its twin 1x1 as two ops
            korder = 11
        op["w_off"], op["bias_off"], op["korder"], op["detect_level"] = w_off, b_off, korder, level
        if fuse_twin is not None:
            ta, tb = fuse_twin
            ops.append(op)
            wlayout.append(dict(korder=8, wkey=wkey, cin=cin_real, cin_pad=cin, cout=n.c, cout_pad=128, k=n.k, K=K, K_pad=K_pad, w_off=w_off, b_off=b_off, kind=kind, act=act,
                                macs=n.h * n.w * n.c * n.k * n.k * cin_real, fused_next=True))
            wlayout.append(dict(korder=12, wkey=(ta.wkey, tb.wkey), cin=n.c, cin_pad=n.c, cout=ta.c + tb.c, cout_pad=128, k=1, K=n.c, K_pad=n.c, w_off=w_off + 128 * K_pad,
                                b_off=b_off + 128, kind=kind, act=act, macs=n.h * n.w * (ta.c + tb.c) * n.c, fused_prev=True))
            w_off += 128 * K_pad + 128 * n.c
            b_off += 256
            return True
        if getattr(src, "virt_up", None) is not None:      # upsample-on-read: part of this concat exists only at half resolution
            un, uoff = src.virt_up
            t = nodes[un.src[0]]
            op["up_buf"], op["up_ld"], op["up_coff"], op["up_c0"], op["up_C"] = t.home, t.ld, t.coff, uoff, un.c
        ops.append(op)
        wlayout.append(dict(korder=korder, wkey=wkey, cin=cin_real, cin_pad=cin, cout=cout, cout_pad=cout_pad, k=n.k, K=K, K_pad=K_pad, w_off=w_off,
                            b_off=b_off, kind=kind, act=act, macs=n.h * n.w * cout * n.k * n.k * cin_real))
        w_off += cout_pad * K_pad
        b_off += cout_pad

    def emit_simple(typ, n, src, k=0, s=1, p=0, out=None):
        out_home, out_ld, out_coff = out if out is not None else (n.home, n.ld, n.coff)
        op = np.zeros((), OP_DTYPE)
        op["type"], op["in_buf"], op["in_ld"], op["in_coff"] = typ, src.home, src.ld, src.coff
        op["H"], op["W"], op["Cin"] = src.h, src.w, src.c
        op["out_buf"], op["out_ld"], op["out_coff"] = out_home, out_ld, out_coff
        op["Ho"], op["Wo"], op["Cout"] = (src.h + 2 * p - k) // s + 1 if typ == 2 else n.h, (src.w + 2 * p - k) // s + 1 if typ == 2 else n.w, src.c
        op["KH"], op["KW"], op["stride"], op["pad"] = k, k, s, p
        op["detect_level"] = -1
        ops.append(op)

    # ---- twin 1x1 branches (every ELAN block starts with two 1x1 convs of the SAME tensor, cfg/deploy/yolov7-w6.yaml:20-21,
    # 79-80, ...): their outputs sit next to each other in the concat buffer, so they run as ONE conv with stacked weights
    # -- the largest activations are read once instead of twice ----
    fused_into = {}     # node idx of the second twin -> first twin
    twins = {}          # first twin idx -> [first, second] in channel order
    by_src = {}
    for n in nodes:
        if n.idx in live and n.kind == "conv" and n.k == 1 and n.s == 1 and n.home is not None:
            by_src.setdefault((n.src[0], n.act, n.home), []).append(n)
    for key, group in by_src.items():
        if len(group) == 2:
            a, b = sorted(group, key=lambda t: t.coff)
            if a.coff + a.c == b.coff and a.ld == b.ld and a.c % 8 == 0:
                tw_first = a if a.idx < b.idx else b
                second = b if tw_first is a else a
                # the fused op is emitted at the position of the EARLIER node; the later one must not be consumed in between
                # by anything that is emitted before it -- true for ELAN (its consumers come after both)
                twins[tw_first.idx] = [a, b]
                fused_into[second.idx] = tw_first.idx

    pending_copies = {}
    for j, cidx, off in extra_copies:
        pending_copies.setdefault(j, []).append((cidx, off))
    # ---- the 64 -> 128 stride-2 layer + the twin 1x1 behind it as one op (csrc/y7t_conv_ws_s2.hip, FUSE): when the twins are the layer's ONLY consumers ----
    fuse_s2 = {}       # idx of the 3x3 node -> (a, b); the twin op is not emitted
    if _lib.switch("Y7T_CONV_WS_S2_FUSE", "1") != "0":      # (measured, round 4: 667 + 328 us as two launches -> 913 us as one; in the pipeline the list -0.15 ms)
        consumers = {}
        for m in nodes:
            if m.idx in live or m.kind == "detect":
                for j in m.src:
                    consumers.setdefault(j, set()).add(m.idx)
        for first_idx, (ta, tb) in twins.items():
            j = ta.src[0]
            t = nodes[j]
            if (t.kind == "conv" and t.k == 3 and t.s == 2 and t.c == 128 and ta.c + tb.c == 128 and ta.act == t.act == tb.act and consumers.get(j) == {ta.idx, tb.idx}
                    and j not in pending_copies and ta.idx not in pending_copies and tb.idx not in pending_copies and nodes[t.src[0]].c == 64
                    and ws_s2_eligible(nodes[t.src[0]].h, nodes[t.src[0]].w, 64, 128, 3, 2, t.p, ta.ld, ta.coff, 0, nodes[t.src[0]].ld, nodes[t.src[0]].coff, max_batch)):
                fuse_s2[j] = (ta, tb)
    fused_twin_done = {min(ta.idx, tb.idx) for ta, tb in fuse_s2.values()}
    for n in nodes:
        if n.idx not in live and n.kind != "detect":
            continue
        if n.kind in ("input", "reorg", "concat"):
            pass
        elif n.kind == "conv":
            if n.idx in fused_into or n.idx in fused_twin_done:
                continue
            if n.idx in fuse_s2:
                ta, tb = fuse_s2[n.idx]
                if emit_conv(n, nodes[n.src[0]], ta.home, ta.ld, ta.coff, 0, ta.c + tb.c, n.act, n.wkey, fuse_twin=(ta, tb)):
                    table[n.home] = (0, 2)           # the tensor between the two layers stays in LDS: its arena buffer (320 x 320 x 128 fp16 per frame) is not needed
                else:
                    fused_twin_done.discard(min(ta.idx, tb.idx))
                    emit_conv(n, nodes[n.src[0]], n.home, n.ld, n.coff, 0, n.c, n.act, n.wkey)
            elif n.idx in twins:
                a, b = twins[n.idx]
                fn = Node("conv", n.src, a.c + b.c, 1, 1, 0, n.act)
                fn.h, fn.w = n.h, n.w
                emit_conv(fn, nodes[n.src[0]], a.home, a.ld, a.coff, 0, a.c + b.c, n.act, (a.wkey, b.wkey))
                for t in (a, b):      # either twin may also sit in a second concat: its copy follows the fused launch
                    for cidx, off in pending_copies.pop(t.idx, []):
                        c = nodes[cidx]
                        emit_simple(2, t, t, 1, 1, 0, out=(c.home, c.ld, off))
            else:
                emit_conv(n, nodes[n.src[0]], n.home, n.ld, n.coff, 0, n.c, n.act, n.wkey)
        elif n.kind == "up":
            if not n.virtual:
                emit_simple(1, n, nodes[n.src[0]])
        elif n.kind == "pool":
            emit_simple(2, n, nodes[n.src[0]], n.k, n.s, n.p)
        elif n.kind == "detect":
            ex = n.extra
            for l, j in enumerate(n.src):
                src = nodes[j]
                hn = Node("conv", [j], ex["na"] * ex["no"], 1, 1, 0, 0)
                hn.h, hn.w = src.h, src.w
                hb = len(table)
                table.append((src.h * src.w * ex["na"] * ex["no"], 4))
                emit_conv(hn, src, hb, ex["na"] * ex["no"], 0, 1, ex["na"] * ex["no"], 0, "model.%d.m.%d" % (n.layer, l), kind=ex["kind"], level=l)
                wlayout[-1]["level"] = l
                heads.append(dict(buf=hb, ny=src.h, nx=src.w, stride=H // src.h))
        for cidx, off in pending_copies.get(n.idx, []):   # tensor that sits in a second concat: copy (1x1 max-pool)
            c = nodes[cidx]
            emit_simple(2, n, n, 1, 1, 0, out=(c.home, c.ld, off))
    plan = Plan()
    plan.ops = np.array(ops, dtype=OP_DTYPE)
    plan.buf_elems = table
    offs, o = [], 0
    for elems, isz in table:
        offs.append(o)
        o = _rup(o + elems * isz * max_batch, 256)
    plan.buf_offsets = np.array(offs, dtype=np.int64)
    plan.arena_bytes = o + 256
    plan.wlayout, plan.w_elems, plan.b_elems = wlayout, w_off, b_off
    plan.heads, plan.det = heads, det.extra
    plan.in_ld, plan.reorg, plan.H, plan.W, plan.max_batch = in_ld, first is not None, H, W, max_batch
    plan.macs = sum(w["macs"] for w in wlayout)
    plan.nodes = nodes
    return plan
