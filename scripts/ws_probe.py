"""What bounds the weights-stationary 64 -> 64 kernel at the benchmarked size?  One layer (H x H, 3x3 / 1), timed with HIP events over `reps` launches, for:
   data      randn / zeros (operand power: the clock the chip holds)
   layout    dense (ld 64 -> 64), in-slice (ld 256 -> 64), out-slice (ld 64 -> 256), both slices (ld 256 -> 256)
   frames    working set inside / outside the 256 MB Infinity Cache
prints us per launch, us per tile ROUND (ceil(tiles / 256) rounds per launch: one persistent workgroup per CU), TFLOP/s."""
import os as _os
_os.environ.setdefault('Y7T_LIB', _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'yolov7-tracker_amd', 'lib', 'liby7t_ablate.so'))      # Y7T_WS_ABLATE instances live in the measuring build
import sys, os, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import _lib
L = _lib.load()
zeros = torch.zeros(256, dtype=torch.float16, device="cuda")
reps = int(os.environ.get("REPS", "30"))


def run(H, B, in_ld, out_ld, data):
    gen = torch.randn if data == "randn" else torch.zeros
    x = gen((B, H, H, in_ld), device="cuda").half()
    w = (gen((64, 576), device="cuda") / 24.0).half()
    b = gen(64, device="cuda").float()
    out = torch.empty((B, H, H, out_ld), device="cuda", dtype=torch.float16)
    def go():
        _lib.check(L.y7t_conv2d_nhwc_f16(_lib.ptr(x), in_ld, 0, B, H, H, 64, _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), out_ld, 0, 0, 64, 64, 3, 3, 1, 1, 1 | 8192,
                                         _lib.ptr(zeros), _lib.stream_ptr()))
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tiles = B * (H // 16) ** 2
    rounds = math.ceil(tiles / 256)
    print("%4dx%-4d B=%-3d ld %3d->%-3d %-5s  %8.1f us  %6.2f us/round  %7.1f TFLOP/s  (%d tiles, %d rounds, %.0f MB in + %.0f MB out touched)"
          % (H, H, B, in_ld, out_ld, data, us, us / rounds, 2.0 * B * H * H * 64 * 576 / us / 1e6, tiles, rounds, B * H * H * 128 / 1e6, B * H * H * 128 / 1e6))
    assert L.y7t_last_kernel().decode().startswith("ws64")
    sys.stdout.flush()


if os.environ.get("QUICK"):       # one line per data kind (ablation sweeps: Y7T_WS_ABLATE=n QUICK=1)
    for data in ("randn", "randn", "zeros"):      # (the first line of a process is a warm-up)
        run(320, 32, 256, 256, data)
    sys.exit(0)
for data in ("randn", "zeros"):
    for (in_ld, out_ld) in ((64, 64), (256, 64), (64, 256), (256, 256)):
        run(320, 32, in_ld, out_ld, data)
for B in (4, 8, 16, 32, 64):
    run(320, B, 64, 64, "randn")
for B in (16, 64, 128):
    run(160, B, 64, 64, "randn")
