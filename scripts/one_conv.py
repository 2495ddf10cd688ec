"""run ONE conv shape repeatedly (for rocprofv3 --pmc): python scripts/one_conv.py H W Cin Cout k s [B reps]"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import _lib
H, W, Cin, Cout, k, s = [int(v) for v in sys.argv[1:7]]
B = int(sys.argv[7]) if len(sys.argv) > 7 else 8
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
L = _lib.load()
pad = k // 2
Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
K = k * k * Cin; K_pad = (K + 63) // 64 * 64; Cout_pad = (Cout + 63) // 64 * 64
x = torch.randn((B, H, W, Cin), device="cuda").half()
w = (torch.randn((Cout_pad, K_pad), device="cuda") / K ** 0.5).half()
b = torch.randn(Cout_pad, device="cuda")
out = torch.empty((B, Ho, Wo, Cout), device="cuda", dtype=torch.float16)
zeros = torch.zeros(256, dtype=torch.float16, device="cuda")
for _ in range(reps):
    _lib.check(L.y7t_conv2d_nhwc_f16(_lib.ptr(x), Cin, 0, B, H, W, Cin, _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), Cout, 0, 0, Cout, Cout_pad, k, k, s, pad, 1,
                                     _lib.ptr(zeros), _lib.stream_ptr()))
torch.cuda.synchronize()
