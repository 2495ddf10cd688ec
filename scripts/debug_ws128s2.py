"""debug: which tap of the stride-2 ws128 kernel reads the wrong input at the left image border (identity filter banks, one tap at a time)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from yolov7_tracker_amd import _lib
from yolov7_tracker_amd.detector import weights
L = _lib.load()
B, H, W, C = 1, 8, 64, 128
rng = np.random.default_rng(0)
x = rng.integers(1, 100, (B, H, W, C)).astype(np.float16)
xd = torch.from_numpy(x).cuda()
zeros = torch.zeros(128, dtype=torch.float16, device="cuda")
bd = torch.zeros(C, device="cuda")
for kh in range(3):
    for kw in range(3):
        Wt = np.zeros((C, C, 3, 3), np.float32)
        Wt[np.arange(C), np.arange(C), kh, kw] = 1.0
        blk = Wt.transpose(0, 2, 3, 1).reshape(C, -1).astype(np.float16)
        wd = torch.from_numpy(weights.pack_ws128(blk)).cuda()
        out = torch.full((B, H // 2, W // 2, C), 7.0, dtype=torch.float16, device="cuda")
        _lib.check(L.y7t_conv2d_nhwc_f16(_lib.ptr(xd), C, 0, B, H, W, C, _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(out), C, 0, 0, C, C, 3, 3, 2, 1, 0 | 16384, _lib.ptr(zeros), _lib.stream_ptr()))
        torch.cuda.synchronize()
        got = out.float().cpu().numpy()
        xp = np.zeros((B, H + 2, W + 2, C), np.float32); xp[:, 1:-1, 1:-1] = x
        ref = xp[:, kh:kh + H:2, kw:kw + W:2]
        bad = got != ref
        idx = np.argwhere(bad)
        msg = ""
        if len(idx):
            r, c, ch = idx[0][1:]
            # where does the wrong value come from?
            src = np.argwhere(x[0] == got[0, r, c, ch])
            src = [tuple(s) for s in src if s[2] == ch][:3]
            msg = "first bad (r %d, c %d, ch %d): got %g want %g; input positions holding that value in this channel: %s" % (r, c, ch, got[0, r, c, ch], ref[0, r, c, ch], src)
        print("tap (%d,%d): bad %d; by col %s %s" % (kh, kw, int(bad.sum()), np.bincount(idx[:, 2], minlength=W // 2)[:4].tolist() if len(idx) else [], msg))
