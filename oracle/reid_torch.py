"""oracle/reid_torch.py -- TEST INFRASTRUCTURE ONLY.

Plain-PyTorch fp32 CPU restatement of the reference's ReID path, the oracle of csrc/y7t_reid.hip:

  * OSNet forward in eval mode      /root/reference/tracker/reid_models/OSNet.py:28-438  (ConvLayer, Conv1x1, Conv1x1Linear, LightConv3x3,
                                    ChannelGate, OSBlock, OSNet.featuremaps / forward -> fc output)
  * crop + Extractor._preprocess    /root/reference/tracker/deepsort.py:28-34, tracker/reid_models/deepsort_reid.py:112-146
                                    (/255, cv2.resize INTER_LINEAR on the float image -- PARITY UNPINNED for cv2 itself: geometry restated
                                    in float32 --, ToTensor, Normalize in the frame's channel order)

  * DeepSORT's own embedding network  /root/reference/tracker/reid_models/deepsort_reid.py:14-110 (BasicBlock, make_layers, Net with reid=True:
                                    conv 3x3 + BN + ReLU + MaxPool(3, 2, 1), four stages of two residual blocks 64/128/256/512, AvgPool (8, 4), x / |x|)
                                    on the `net_dict` state dict of its checkpoint (weights/ckpt.t7, which the reference does not ship)

It works on the same torchreid-style state dict as the product and is pinned against the reference's own OSNet class (random weights and
weights/osnet_x0_25.pth) in tests/test_reid_oracle.py where /root/reference exists."""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-5


def _bn(x, sd, name):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"], False, 0.0, EPS)


def _conv_layer(x, sd, name, stride=1, padding=0, relu=True):
    x = _bn(F.conv2d(x, sd[name + ".conv.weight"], None, stride, padding), sd, name + ".bn")
    return F.relu(x) if relu else x


def _light(x, sd, name):
    x = F.conv2d(x, sd[name + ".conv1.weight"])
    x = F.conv2d(x, sd[name + ".conv2.weight"], None, 1, 1, 1, x.shape[1])
    return F.relu(_bn(x, sd, name + ".bn"))


def _gate(x, sd, name):
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.relu(F.conv2d(g, sd[name + ".fc1.weight"], sd[name + ".fc1.bias"]))
    g = torch.sigmoid(F.conv2d(g, sd[name + ".fc2.weight"], sd[name + ".fc2.bias"]))
    return x * g


def _osblock(x, sd, name):
    x1 = _conv_layer(x, sd, name + ".conv1")
    a = _light(x1, sd, name + ".conv2a")
    outs = [a]
    for tag, n in (("conv2b", 2), ("conv2c", 3), ("conv2d", 4)):
        t = x1
        for j in range(n):
            t = _light(t, sd, "%s.%s.%d" % (name, tag, j))
        outs.append(t)
    x2 = sum(_gate(o, sd, name + ".gate") for o in outs)
    x3 = _conv_layer(x2, sd, name + ".conv3", relu=False)
    idn = _conv_layer(x, sd, name + ".downsample", relu=False) if name + ".downsample.conv.weight" in sd else x
    return F.relu(x3 + idn)


@torch.no_grad()
def osnet_forward(sd, x, layers=(2, 2, 2)):
    """x: (N, 3, H, W) float32 -> (N, feature_dim) (OSNet.forward, eval mode)"""
    sd = {k: v.float() for k, v in sd.items()}
    x = _conv_layer(x.float(), sd, "conv1", 2, 3)
    x = F.max_pool2d(x, 3, 2, 1)
    for si, nblk in enumerate(layers):
        stage = "conv%d" % (si + 2)
        for bi in range(nblk):
            x = _osblock(x, sd, "%s.%d" % (stage, bi))
        if si < 2:
            x = F.avg_pool2d(_conv_layer(x, sd, "%s.%d.0" % (stage, nblk)), 2, 2)
    x = _conv_layer(x, sd, "conv5")
    v = F.adaptive_avg_pool2d(x, 1).flatten(1)
    v = F.linear(v, sd["fc.0.weight"], sd["fc.0.bias"])
    v = F.batch_norm(v, sd["fc.1.running_mean"], sd["fc.1.running_var"], sd["fc.1.weight"], sd["fc.1.bias"], False, 0.0, EPS)
    return F.relu(v)


def _basic_block(x, sd, name, stride):
    """deepsort_reid.py:14-49"""
    y = F.relu(_bn(F.conv2d(x, sd[name + ".conv1.weight"], None, stride, 1), sd, name + ".bn1"))
    y = _bn(F.conv2d(y, sd[name + ".conv2.weight"], None, 1, 1), sd, name + ".bn2")
    if name + ".downsample.0.weight" in sd:
        x = _bn(F.conv2d(x, sd[name + ".downsample.0.weight"], None, stride), sd, name + ".downsample.1")
    return F.relu(x.add(y))


def deepsort_net_forward(sd, x):
    """Net.forward with reid=True in eval mode (deepsort_reid.py:62-110): (N, 3, 128, 64) -> (N, 512) unit vectors"""
    x = F.relu(_bn(F.conv2d(x, sd["conv.0.weight"], sd["conv.0.bias"], 1, 1), sd, "conv.1"))
    x = F.max_pool2d(x, 3, 2, padding=1)
    for li, down in ((1, False), (2, True), (3, True), (4, True)):
        for bi in range(2):
            x = _basic_block(x, sd, "layer%d.%d" % (li, bi), 2 if (down and bi == 0) else 1)
    x = F.avg_pool2d(x, (8, 4), 1)
    x = x.view(x.size(0), -1)
    return x.div(x.norm(p=2, dim=1, keepdim=True))


def resize_linear_f32(img, new_h, new_w):
    """cv2.resize(float image, (new_w, new_h)) INTER_LINEAR geometry in float32"""
    H0, W0 = img.shape[:2]
    f32 = np.float32
    fy = (np.arange(new_h, dtype=f32) + f32(0.5)) * f32(f32(H0) / f32(new_h)) - f32(0.5)
    fx = (np.arange(new_w, dtype=f32) + f32(0.5)) * f32(f32(W0) / f32(new_w)) - f32(0.5)
    y0, x0 = np.floor(fy).astype(np.int64), np.floor(fx).astype(np.int64)
    wy, wx = (fy - y0.astype(f32))[:, None, None], (fx - x0.astype(f32))[None, :, None]
    y1, x1 = np.clip(y0 + 1, 0, H0 - 1), np.clip(x0 + 1, 0, W0 - 1)
    y0, x0 = np.clip(y0, 0, H0 - 1), np.clip(x0, 0, W0 - 1)
    t0 = (f32(1) - wx) * img[y0][:, x0] + wx * img[y0][:, x1]
    t1 = (f32(1) - wx) * img[y1][:, x0] + wx * img[y1][:, x1]
    return ((f32(1) - wy) * t0 + wy * t1).astype(f32)


def preprocess(frame_bgr_u8, tlbrs, size=(64, 128)):
    """deepsort.py:28-34 crops + Extractor._preprocess -> (N, 3, H, W) float32"""
    mean, std = np.float32([0.485, 0.456, 0.406]), np.float32([0.229, 0.224, 0.225])
    out = []
    for b in tlbrs:
        x1, y1, x2, y2 = (int(v) for v in b)
        crop = frame_bgr_u8[y1:y2, x1:x2].astype(np.float32) / np.float32(255.0)
        r = resize_linear_f32(crop, size[1], size[0])
        out.append(((r - mean) / std).transpose(2, 0, 1))
    return torch.from_numpy(np.stack(out)) if out else torch.zeros((0, 3, size[1], size[0]))
