"""YOLOv7 graph specifications in the reference's layer-list form `[from, number, module, args]`
(/root/reference/models/yolo.py:443-520 consumes exactly this from cfg/*.yaml).

`yolov7_w6(nc)` / `yolov7_tiny(nc)` generate the lists programmatically (they are checked against the reference's
cfg/deploy/yolov7-w6.yaml and cfg/deploy/yolov7-tiny.yaml by tests/test_detector_graph.py where the reference is
present); `load_yaml(path)` reads any cfg file of the same format (the CLI's --model_cfg)."""
import yaml

W6_ANCHORS = [[19, 27, 44, 40, 38, 94], [96, 68, 86, 152, 180, 137], [140, 301, 303, 264, 238, 542], [436, 615, 739, 380, 925, 792]]
TINY_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


def _conv(f, c, k=1, s=1, act=None):
    return [f, 1, "Conv", [c, k, s] if act is None else [c, k, s, None, 1, act]]


def yolov7_w6(nc=80, training=False):
    L = [[-1, 1, "ReOrg", []], _conv(-1, 64, 3, 1)]
    for c in (128, 256, 512, 768, 1024):            # five down-stages: 3x3/s2 + ELAN(4-way concat) + 1x1
        h = c // 2
        L += [_conv(-1, c, 3, 2), _conv(-1, h), _conv(-2, h)] + [_conv(-1, h, 3, 1) for _ in range(4)]
        L += [[[-1, -3, -5, -6], 1, "Concat", [1]], _conv(-1, c)]
    L.append([-1, 1, "SPPCSPC", [512]])              # 47

    def elan_h(c):                                   # head block: two 1x1, four chained 3x3 (c/2), 6-way concat, 1x1
        return [_conv(-1, c), _conv(-2, c)] + [_conv(-1, c // 2, 3, 1) for _ in range(4)] + \
            [[[-1, -2, -3, -4, -5, -6], 1, "Concat", [1]], _conv(-1, c)]
    for c, route in ((384, 37), (256, 28), (128, 19)):   # top-down
        L += [_conv(-1, c), [-1, 1, "nn.Upsample", [None, 2, "nearest"]], _conv(route, c), [[-1, -2], 1, "Concat", [1]]] + elan_h(c)
    for c, route in ((256, 71), (384, 59), (512, 47)):   # bottom-up
        L += [_conv(-1, c, 3, 2), [[-1, route], 1, "Concat", [1]]] + elan_h(c)
    L += [_conv(83, 256, 3, 1), _conv(93, 512, 3, 1), _conv(103, 768, 3, 1), _conv(113, 1024, 3, 1)]
    if training:       # cfg/training/yolov7-w6.yaml:156-162 -- what train_aux.py produces: four aux-head convs + IAuxDetect (models/yolo.py:111-158)
        L += [_conv(83, 320, 3, 1), _conv(71, 640, 3, 1), _conv(59, 960, 3, 1), _conv(47, 1280, 3, 1)]
        L.append([[114, 115, 116, 117, 118, 119, 120, 121], 1, "IAuxDetect", ["nc", "anchors"]])
    else:
        L.append([[114, 115, 116, 117], 1, "Detect", ["nc", "anchors"]])
    return {"nc": nc, "depth_multiple": 1.0, "width_multiple": 1.0, "anchors": W6_ANCHORS, "layers": L, "n_backbone": 47}


def yolov7_w6_training(nc=80):
    """the graph of the checkpoints the reference's training actually saves for w6 (README.md:101, cfg/training/yolov7-w6.yaml): at inference the aux
    branch is dead work (yolo.py:141-153 computes and discards it) and ImplicitA / ImplicitM fold into the main head's 1x1 convs"""
    return yolov7_w6(nc, training=True)


def yolov7_tiny(nc=80):
    A = "nn.LeakyReLU(0.1)"

    def c(f, ch, k=1, s=1):
        return _conv(f, ch, k, s, A)

    def elan_t(h, out):
        return [c(-1, h), c(-2, h), c(-1, h, 3, 1), c(-1, h, 3, 1), [[-1, -2, -3, -4], 1, "Concat", [1]], c(-1, out)]
    L = [c(-1, 32, 3, 2), c(-1, 64, 3, 2)] + elan_t(32, 64)
    for h, out in ((64, 128), (128, 256), (256, 512)):
        L += [[-1, 1, "MP", []]] + elan_t(h, out)
    L += [c(-1, 256), c(-2, 256), [-1, 1, "SP", [5]], [-2, 1, "SP", [9]], [-3, 1, "SP", [13]], [[-1, -2, -3, -4], 1, "Concat", [1]],
          c(-1, 256), [[-1, -7], 1, "Concat", [1]], c(-1, 256)]                                     # 29..37
    for ch, route, h in ((128, 21, 64), (64, 14, 32)):
        L += [c(-1, ch), [-1, 1, "nn.Upsample", [None, 2, "nearest"]], c(route, ch), [[-1, -2], 1, "Concat", [1]]] + elan_t(h, ch)
    for ch, route, h in ((128, 47, 64), (256, 37, 128)):
        L += [c(-1, ch, 3, 2), [[-1, route], 1, "Concat", [1]]] + elan_t(h, ch)
    L += [c(57, 128, 3, 1), c(65, 256, 3, 1), c(73, 512, 3, 1), [[74, 75, 76], 1, "Detect", ["nc", "anchors"]]]
    return {"nc": nc, "depth_multiple": 1.0, "width_multiple": 1.0, "anchors": TINY_ANCHORS, "layers": L, "n_backbone": 29}


def load_yaml(path, nc=None):
    with open(path) as f:
        d = yaml.safe_load(f)
    spec = {"nc": d["nc"] if nc is None else nc, "depth_multiple": d.get("depth_multiple", 1.0), "width_multiple": d.get("width_multiple", 1.0),
            "anchors": d["anchors"], "layers": list(d["backbone"]) + list(d["head"]), "n_backbone": len(d["backbone"])}
    return spec


ARCHS = {"yolov7-w6": yolov7_w6, "yolov7-tiny": yolov7_tiny, "yolov7-w6-training": yolov7_w6_training}
