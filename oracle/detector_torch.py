"""oracle/detector_torch.py -- TEST INFRASTRUCTURE ONLY.

Plain-PyTorch fp32 CPU restatement of the reference's detector path, used as the floating-point oracle of the HIP
convolution / decode / NMS kernels on machines where /root/reference does not exist:

  * forward walk      /root/reference/models/yolo.py:321-351 (forward_once), layer semantics models/common.py:23-111
                      (autopad, MP, SP, ReOrg, Concat, Conv = act(bn(conv(x)))), :262-280 (SPPCSPC), yolo.py:39-57 (Detect)
  * NMS               utils/general.py:607-695 with torchvision.ops.nms restated by oracle/y7t_oracle.c (greedy, strict >)
  * scale_coords      utils/general.py:319-340 and the .round() of tracker/track.py:240

It consumes the same graph description (yolov7_tracker_amd.detector.graph.parse -- host logic that carries no
arithmetic) and the same reference-style state dict as the product.  Pinned against the reference's own
models.yolo.Model / utils.general.non_max_suppression in tests/test_detector_oracle.py (build container only).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cnative

BN_EPS = 1e-3


def _to_f16(w64):
    """float64 -> the nearest fp16 value (ONE rounding, numpy's conversion), as float32.  torch's double.half() goes through float32 and
    double-rounds: about one folded weight in 10^4 lands on the other neighbour, a 1e-3 relative change of that weight."""
    return torch.from_numpy(w64.numpy().astype(np.float16).astype(np.float32))


def _conv_bn_act(x, sd, key, k, s, p, act, fp16=False, round_out=True):
    if fp16:
        # storage-precision emulation of the HIP path: BN folded into the weights (utils/torch_utils.py:181-201), folded
        # weights rounded to fp16, fp32 accumulate + fp32 bias + activation, result rounded to fp16
        w = sd[key + ".conv.weight"].double()
        b = sd[key + ".conv.bias"].double() if key + ".conv.bias" in sd else torch.zeros(w.shape[0], dtype=torch.float64)
        if key + ".bn.weight" in sd:
            scale = sd[key + ".bn.weight"].double() / torch.sqrt(sd[key + ".bn.running_var"].double() + BN_EPS)
            w = w * scale[:, None, None, None]
            b = (b - sd[key + ".bn.running_mean"].double()) * scale + sd[key + ".bn.bias"].double()
        y = F.conv2d(x, _to_f16(w), b.float(), stride=s, padding=p)
        y = F.silu(y) if act == 1 else (F.leaky_relu(y, 0.1) if act == 2 else y)
        return y.half().float() if round_out else y      # round_out=False: the exact fp32 value the fp16 store of the HIP path rounds
    w = sd[key + ".conv.weight"].float()
    b = sd.get(key + ".conv.bias")
    y = F.conv2d(x, w, None if b is None else b.float(), stride=s, padding=p)
    if key + ".bn.weight" in sd:
        y = F.batch_norm(y, sd[key + ".bn.running_mean"].float(), sd[key + ".bn.running_var"].float(), sd[key + ".bn.weight"].float(),
                         sd[key + ".bn.bias"].float(), False, 0.0, BN_EPS)
    if act == 1:
        y = F.silu(y)
    elif act == 2:
        y = F.leaky_relu(y, 0.1)
    return y


def conv_abs_sum(x, sd, key, s, p):
    """sum_k |w_k x_k| + |b| per output of the BN-folded fp16-weight conv `key` (pre-activation): the scale of the forward error bound of
    its fp32 accumulation, |fl(sum) - sum| <= c u sum|w_k x_k| (Higham, Accuracy and Stability of Numerical Algorithms, section 4.2)"""
    w = sd[key + ".conv.weight"].double()
    b = sd[key + ".conv.bias"].double() if key + ".conv.bias" in sd else torch.zeros(w.shape[0], dtype=torch.float64)
    if key + ".bn.weight" in sd:
        scale = sd[key + ".bn.weight"].double() / torch.sqrt(sd[key + ".bn.running_var"].double() + BN_EPS)
        w = w * scale[:, None, None, None]
        b = (b - sd[key + ".bn.running_mean"].double()) * scale + sd[key + ".bn.bias"].double()
    return F.conv2d(x.abs(), _to_f16(w).abs(), b.float().abs(), stride=s, padding=p)


@torch.no_grad()
def forward(nodes, sd, img, anchors, keep=False, fp16=False):
    """img (B,3,H,W) float32 -> (decoded (B,A,no), raw list of (B,na,ny,nx,no), {node idx: tensor} if keep).
    fp16=True emulates the storage precision of the HIP path (fp16 weights/activations, fp32 accumulate)."""
    vals = {0: img.float().half().float() if fp16 else img.float()}
    raw, z = [], []
    H = img.shape[2]
    for n in nodes[1:]:
        src = [vals[j] for j in n.src] if n.kind != "detect" else None
        if n.kind == "reorg":
            x = src[0]
            y = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
        elif n.kind == "conv":
            if any(j not in vals for j in n.src):
                continue
            y = _conv_bn_act(src[0], sd, n.wkey, n.k, n.s, n.p, n.act, fp16)
        elif n.kind == "concat":
            y = torch.cat(src, 1)
        elif n.kind == "up":
            y = F.interpolate(src[0], scale_factor=2, mode="nearest")
        elif n.kind == "pool":
            y = F.max_pool2d(src[0], n.k, n.s, n.p)
        elif n.kind == "detect":
            ex = n.extra
            a = torch.tensor(anchors, dtype=torch.float32).view(ex["nl"], -1, 2)
            for l, j in enumerate(n.src):
                x = vals[j]
                base = "model.%d" % n.layer
                if ex["kind"] in ("IDetect", "IAuxDetect") and "%s.ia.%d.implicit" % (base, l) in sd:
                    x = x + sd["%s.ia.%d.implicit" % (base, l)].float()
                wd = sd["%s.m.%d.weight" % (base, l)].float()
                x = F.conv2d(x, wd.half().float() if fp16 else wd, sd["%s.m.%d.bias" % (base, l)].float())
                if ex["kind"] in ("IDetect", "IAuxDetect") and "%s.im.%d.implicit" % (base, l) in sd:
                    x = x * sd["%s.im.%d.implicit" % (base, l)].float()
                bs, _, ny, nx = x.shape
                x = x.view(bs, ex["na"], ex["no"], ny, nx).permute(0, 1, 3, 4, 2).contiguous()
                raw.append(x)
            z = [decode_level(x, a[l], H / x.shape[2]) for l, x in enumerate(raw)]
            continue
        else:
            raise NotImplementedError(n.kind)
        vals[n.idx] = y
    out = (torch.cat(z, 1), raw)
    return out + (vals,) if keep else out


def decode_level(x, anchors_l, stride):
    """Detect.forward inference branch for one level (models/yolo.py:49-56): x (bs, na, ny, nx, no) raw conv output -> (bs, na*ny*nx, no)"""
    bs, na, ny, nx, no = x.shape
    yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
    grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()          # _make_grid, yolo.py:59-62
    y = x.float().sigmoid()
    y[..., 0:2] = (y[..., 0:2] * 2. - 0.5 + grid) * stride
    y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * anchors_l.view(1, na, 1, 1, 2)
    return y.view(bs, -1, no)


def decode_heads(raw, anchors, H):
    """the reference's `model(img)[0]` from the raw Detect conv outputs: cat over levels (yolo.py:57)"""
    a = torch.tensor(anchors, dtype=torch.float32).view(len(raw), -1, 2)
    return torch.cat([decode_level(x, a[l], H / x.shape[2]) for l, x in enumerate(raw)], 1)


def xywh2xyxy(x):
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


@torch.no_grad()
def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, nms_fn=None):
    """utils/general.py:607-695 with classes=None, agnostic=False, multi_label=False, labels=()."""
    nms_fn = nms_fn or (lambda b, s, t: torch.from_numpy(cnative.nms(b.numpy(), s.numpy(), t)))
    xc = prediction[..., 4] > conf_thres
    max_wh, max_det, max_nms = 4096, 300, 30000
    output = [torch.zeros((0, 6))] * prediction.shape[0]
    for xi, x in enumerate(prediction):
        x = x[xc[xi]]
        if not x.shape[0]:
            continue
        x = x.clone()
        x[:, 5:] *= x[:, 4:5]
        box = xywh2xyxy(x[:, :4])
        conf, j = x[:, 5:].max(1, keepdim=True)
        x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        n = x.shape[0]
        if not n:
            continue
        elif n > max_nms:
            x = x[x[:, 4].argsort(descending=True)[:max_nms]]
        c = x[:, 5:6] * max_wh
        boxes, scores = x[:, :4] + c, x[:, 4]
        i = nms_fn(boxes, scores, iou_thres)
        if i.shape[0] > max_det:
            i = i[:max_det]
        output[xi] = x[i]
    return output


def scale_coords_round(img1_shape, coords, img0_shape):
    """scale_coords (general.py:319-340) followed by .round() (tracker/track.py:240); coords (n,4) float32 tensor"""
    coords = coords.clone()
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    coords[:, 0].clamp_(0, img0_shape[1])
    coords[:, 1].clamp_(0, img0_shape[0])
    coords[:, 2].clamp_(0, img0_shape[1])
    coords[:, 3].clamp_(0, img0_shape[0])
    return coords.round()


def candidates(dec, conf_thres=0.01):
    """the candidate filter of non_max_suppression (utils/general.py:629-662) on ONE image's decoded tensor (A, 5+nc), without the NMS:
    -> dict anchor row -> (xyxy float32[4], conf, cls, per-class conf vector) for rows with obj > conf_thres and best-class conf > conf_thres"""
    x = dec.clone()
    rows = torch.nonzero(x[:, 4] > conf_thres).flatten()
    x = x[rows]
    x[:, 5:] *= x[:, 4:5]
    box = x[:, :4].clone()
    box[:, 0], box[:, 1] = x[:, 0] - x[:, 2] / 2, x[:, 1] - x[:, 3] / 2
    box[:, 2], box[:, 3] = x[:, 0] + x[:, 2] / 2, x[:, 1] + x[:, 3] / 2
    conf, j = x[:, 5:].max(1)
    ok = conf > conf_thres
    return {int(r): (b.numpy(), float(c), int(k), sc.numpy()) for r, b, c, k, sc in zip(rows[ok], box[ok], conf[ok], j[ok], x[:, 5:][ok])}


def box_iou_1(a, b):
    """IoU of two xyxy boxes (plain area convention, no +1: the convention of torchvision.ops.box_iou / utils/general.py:box_iou)"""
    iw = max(0.0, min(float(a[2]), float(b[2])) - max(float(a[0]), float(b[0])))
    ih = max(0.0, min(float(a[3]), float(b[3])) - max(float(a[1]), float(b[1])))
    ua = (float(a[2]) - float(a[0])) * (float(a[3]) - float(a[1])) + (float(b[2]) - float(b[0])) * (float(b[3]) - float(b[1])) - iw * ih
    return iw * ih / ua if ua > 0 else 1.0


def compare_candidate_sets(got, want, conf_thres=0.01, px=1.0, dconf=5e-3, iou=None):
    """got / want: dicts anchor row -> (xyxy, conf, cls) of ONE image (device / oracle).  SURVEY 8a's bar applied BEFORE the NMS, where no greedy
    order can amplify a rounding difference: every candidate both sides have must agree in class (or pick a class the oracle scores within dconf of its best), |dcoord| <= px, |dconf| <= dconf; a candidate
    only one side has must sit within dconf of the threshold (it crossed conf_thres by rounding noise).  iou (e.g. 0.99): 8a's FULL coordinate bar --
    "IoU >= 0.99 / |dcoord| <= 1 px": a box passes on either (large boxes of the coarse Detect levels pass on IoU, small ones on pixels).  -> statistics dict"""
    both = sorted(set(got) & set(want))
    only = sorted(set(got) ^ set(want))
    dc = np.array([np.abs(got[r][0] - want[r][0]).max() for r in both]) if both else np.zeros(0)
    ds = np.array([abs(got[r][1] - want[r][1]) for r in both]) if both else np.zeros(0)
    side = np.array([max(want[r][0][2] - want[r][0][0], want[r][0][3] - want[r][0][1]) for r in both]) if both else np.zeros(0)
    # a different class is a difference only if the oracle does not score the device's class within dconf of its own best (two classes tied within the
    # conf tolerance: `conf, j = x[:, 5:].max(1)` picks by rounding noise)
    cls_diff = [r for r in both if got[r][2] != want[r][2] and (len(want[r]) < 4 or want[r][3][got[r][2]] < want[r][1] - dconf)]
    only_margin = np.array([abs((got.get(r) or want.get(r))[1] - conf_thres) for r in only]) if only else np.zeros(0)
    coord_ok = dc <= px
    st = {}
    if iou is not None:
        ious = np.array([box_iou_1(got[r][0], want[r][0]) for r in both]) if both else np.zeros(0)
        coord_ok = coord_ok | (ious >= iou)
        wh = np.array([[want[r][0][2] - want[r][0][0], want[r][0][3] - want[r][0][1]] for r in both]) if both else np.zeros((0, 2))
        st = {"out_of_coord_bar": [{"row": int(both[i]), "dcoord": float(dc[i]), "iou": float(ious[i]), "w": float(wh[i][0]), "h": float(wh[i][1])} for i in np.nonzero(~coord_ok)[0]],
              "min_iou": float(ious.min()) if len(ious) else 1.0, "min_iou_of_boxes_off_by_more_than_px": float(ious[dc > px].min()) if bool((dc > px).any()) else 1.0,
              "n_pass_on_iou_only": int(((dc > px) & (ious >= iou)).sum()), "max_side_px": float(side.max()) if len(side) else 0.0}
    st.update({"n_got": len(got), "n_want": len(want), "n_both": len(both), "n_only_one_side": len(only),
               "max_dcoord": float(dc.max()) if len(dc) else 0.0, "max_dconf": float(ds.max()) if len(ds) else 0.0,
               "max_dcoord_rel_side": float((dc / np.maximum(side, 1.0)).max()) if len(dc) else 0.0,
               "frac_within_bar": float((coord_ok & (ds <= dconf)).mean()) if len(dc) else 1.0,
               "n_class_differs": len(cls_diff), "max_margin_only_one_side": float(only_margin.max()) if len(only) else 0.0,
               "worst_rows": [both[i] for i in np.argsort(-dc)[:3]] if len(dc) else []})
    return st


def nms_rows(cands, iou_thres=0.45, max_nms=30000, max_det=300, max_wh=4096):
    """the NMS half of non_max_suppression (utils/general.py:664-695: per-class offset boxes, max_nms best candidates, greedy torchvision.ops.nms restated in
    oracle/y7t_oracle.c, first max_det) on ONE image's candidate dict anchor row -> (xyxy, conf, cls, ...) -> the kept anchor rows, in output order"""
    rows = np.array(sorted(cands))
    if not len(rows):
        return rows
    box = np.stack([cands[r][0] for r in rows]).astype(np.float32)
    s = np.array([cands[r][1] for r in rows], np.float32)
    c = np.array([cands[r][2] for r in rows], np.float32)
    order = np.lexsort((rows, -s.astype(np.float64)))[:max_nms]
    k = cnative.nms((box + c[:, None] * np.float32(max_wh)).astype(np.float32)[order], s[order], iou_thres)[:max_det]
    return rows[order[k]]


def explain_kept_set_difference(cands_x, kept_x, cands_y, kept_y, score_noise, iou_noise=0.02, conf_thres=0.01, iou_thres=0.45, max_det=300, depth=12):
    """Two runs of the same greedy NMS on two candidate sets that differ by rounding noise (x, y: dicts anchor row -> (xyxy, conf, cls, ...); kept_*: kept rows in
    output order).  For every row kept on ONE side only, walk the greedy decisions that made it so and name the decision that could flip within the noise:
      'score-tie'     the row and the rival that suppresses it on the other side swap order: their scores differ by <= 2 score_noise on the side where the row survives
      'iou-threshold' the rival's IoU with the row crosses iou_thres by less than iou_noise + what the pair's own coordinate deviation can move it (4 d / shortest side)
      'conf-threshold' the row (or a rival up the chain) is a candidate on one side only, its score within score_noise of conf_thres
      'cut'           the row ranks just behind the max_det-th survivor: within 2 score_noise of the last kept score
      'class-tie'     the rival's best class differs between the sides (two class scores within 2 score_noise; needs the per-class vector as 4th tuple entry)
    A rival that is itself kept on one side only is followed recursively (a flip propagates down a chain of overlapping boxes).  -> {row: reason or None}; None
    = a difference that rounding noise of the stated size does NOT explain (an arithmetic difference of the kernels)."""
    f32 = np.float32

    def order_key(c, r):
        return (-float(f32(c[r][1])), r)

    def iou(c, a, b):
        """IoU as the NMS computes it (oracle/y7t_oracle.c:281-305 = torchvision.ops.nms): float32 arithmetic on the class-offset boxes (+ cls * max_wh,
        utils/general.py:675) -- near the threshold that rounding decides, and the explanation must look at the number the decision was made on"""
        A = (np.asarray(c[a][0], f32) + f32(c[a][2]) * f32(4096)).astype(f32)
        B = (np.asarray(c[b][0], f32) + f32(c[b][2]) * f32(4096)).astype(f32)
        w = max(f32(0), f32(min(A[2], B[2]) - max(A[0], B[0])))
        h = max(f32(0), f32(min(A[3], B[3]) - max(A[1], B[1])))
        inter = f32(w * h)
        ua = f32(f32(f32(A[2] - A[0]) * f32(A[3] - A[1])) + f32(f32(B[2] - B[0]) * f32(B[3] - B[1]))) - inter
        return float(inter / ua) if ua > 0 else 0.0

    def class_tie(c, r, k1, k2):
        return len(c[r]) > 3 and abs(float(c[r][3][k1]) - float(c[r][3][k2])) <= 2 * score_noise

    def explain(a, cx, kx, cy, ky, d):
        # `a` is kept on side y and missing on side x
        if d > depth:
            return None
        if a not in cx:
            return "conf-threshold" if abs(cy[a][1] - conf_thres) <= score_noise else None
        if cx[a][2] != cy[a][2]:                       # the row itself changed class between the sides (it competes in another class group there)
            return "class-tie" if (class_tie(cy, a, cx[a][2], cy[a][2]) or class_tie(cx, a, cx[a][2], cy[a][2])) else None
        kxs = list(kx)
        sup = [b for b in kxs if b != a and cx[b][2] == cx[a][2] and order_key(cx, b) < order_key(cx, a) and iou(cx, a, b) > iou_thres]
        if not sup:
            if len(kxs) >= max_det and abs(cx[a][1] - cx[kxs[-1]][1]) <= 2 * score_noise:
                return "cut"
            return None
        b = sup[0]
        if b not in cy:
            return "conf-threshold" if abs(cx[b][1] - conf_thres) <= score_noise else None
        if cy[b][2] != cy[a][2]:                      # the rival changed class between the sides: two of its class scores tie (`conf, j = x[:, 5:].max(1)`)
            return "class-tie" if (class_tie(cy, b, cx[b][2], cy[b][2]) or class_tie(cx, b, cx[b][2], cy[b][2])) else None
        iy, ix = iou(cy, a, b), iou(cx, a, b)
        if iy <= iou_thres:
            # how far the two sides' coordinates of THIS pair can move its IoU: an edge shift d changes the overlap by <= d / (shortest side) per edge -- the random
            # head also emits slivers (0.03 px wide, 130 px tall) whose IoU with a neighbour goes from 0.77 to 0.25 on a 0.016 px shift
            dxy = max(float(np.abs(np.asarray(cx[r][0], np.float64) - np.asarray(cy[r][0], np.float64)).max()) for r in (a, b))
            ms = min(min(float(c[r][0][2] - c[r][0][0]), float(c[r][0][3] - c[r][0][1])) for c in (cx, cy) for r in (a, b))
            tol = iou_noise + 4.0 * (dxy + 0.004) / max(ms, 1e-6)          # + 2 float32 ulps of a class-offset coordinate (28672 + x: 2^-9 px)
            return "iou-threshold" if (iou_thres - iy) <= tol and (ix - iou_thres) <= tol else None
        if order_key(cy, a) < order_key(cy, b):        # on y the row comes first (and, kept, suppresses b there): the two swapped order
            return "score-tie" if abs(cy[a][1] - cy[b][1]) <= 2 * score_noise and abs(cx[a][1] - cx[b][1]) <= 2 * score_noise else None
        if b in set(ky):
            return None                                   # b precedes a on y, overlaps it, is kept -- and a is kept too: not a greedy NMS outcome
        r = explain(b, cy, ky, cx, kx, d + 1)            # b is kept on x, missing on y: the flip happened further up the chain
        return ("chain:" + r) if r else None

    sx, sy = set(kept_x), set(kept_y)
    out = {}
    for a in kept_y:
        if a not in sx:
            out[int(a)] = explain(a, cands_x, kept_x, cands_y, kept_y, 0)
    for a in kept_x:
        if a not in sy:
            out[int(a)] = explain(a, cands_y, kept_y, cands_x, kept_x, 0)
    return out
