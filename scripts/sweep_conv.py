"""time ONE conv shape over a list of batch sizes (tile-quantisation experiments):
python scripts/sweep_conv.py H W Cin Cout k s B1,B2,... [reps]
ACT_BITS=<n> in the environment: the weight-packing bits of y7t_conv2d_nhwc_f16's `act` (256 (kh, chunk, kw) order, 1024 LDS-patch panels, 2048 1x1 panels,
4096 stride-2 patch panels) -- the values are random, only the kernel the dispatcher picks matters here."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from yolov7_tracker_amd import _lib
H, W, Cin, Cout, k, s = [int(v) for v in sys.argv[1:7]]
Bs = [int(v) for v in sys.argv[7].split(",")]
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
L = _lib.load()
pad = k // 2
Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
BITS = int(os.environ.get("ACT_BITS", "0"))
K = k * k * Cin; K_pad = (K + 63) // 64 * 64; Cout_pad = (Cout + 127) // 128 * 128 if BITS & 4096 else (Cout + 63) // 64 * 64
zeros = torch.zeros(256, dtype=torch.float16, device="cuda")
for B in Bs:
    # DATA=zeros|ones|small|randn: operand values (the MFMA power draw -- and with it the sustained clock -- depends on them)
    mode = os.environ.get("DATA", "randn")
    gen = {"zeros": lambda s: torch.zeros(s, device="cuda"), "ones": lambda s: torch.ones(s, device="cuda"),
           "small": lambda s: torch.randint(0, 2, s, device="cuda").float() * 0.5, "randn": lambda s: torch.randn(s, device="cuda")}[mode]
    x = gen((B, H, W, Cin)).half()
    w = (gen((Cout_pad, K_pad)) / K ** 0.5).half()
    b = torch.randn(Cout_pad, device="cuda")
    out = torch.empty((B, Ho, Wo, Cout), device="cuda", dtype=torch.float16)
    def run():
        _lib.check(L.y7t_conv2d_nhwc_f16(_lib.ptr(x), Cin, 0, B, H, W, Cin, _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), Cout, 0, 0, Cout, Cout_pad, k, k, s, pad, 1 | BITS,
                                         _lib.ptr(zeros), _lib.stream_ptr()))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("B=%3d  %8.1f us  %7.1f TF/s" % (B, us, 2.0 * B * Ho * Wo * Cout * K / us / 1e6), flush=True)
