"""per-layer micro-benchmark of the conv launch list (w6 @ 1280, B frames): time every distinct conv shape in isolation
(HIP events, N reps), print TFLOP/s and algorithmic GB/s."""
import sys, ctypes, collections
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import _lib
from yolov7_tracker_amd.detector import arch, graph
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
import os
KORDER = int(os.environ.get('KORDER', '1'))
ACT = int(os.environ.get('ACT', '1'))      # activation of the timed layers (1 SiLU as in the network, 0 none, 2 LeakyReLU): what the epilogue's activation costs
L = _lib.load()
plan = graph.lower(graph.parse(arch.yolov7_w6(10))[0], 1280, 1280, B)
shapes = collections.OrderedDict()
for op in plan.ops:
    if op["type"] != 0: continue
    key = tuple(int(op[k]) for k in ("H", "W", "Cin", "Cout", "Cout_pad", "KH", "stride", "pad", "out_f32", "in_ld", "out_ld"))
    shapes.setdefault(key, 0); shapes[key] += 1
zeros = torch.zeros(256, dtype=torch.float16, device="cuda")
tot_t = tot_f = 0.0
print("%-34s %3s %9s %8s %8s %8s" % ("shape (HxW Cin->Cout k/s)", "x", "us", "TF/s", "GB/s", "blocks"))
ONLY = os.environ.get("ONLY")      # "H,Cin,Cout,k,s": that shape only (long or profiled runs of one layer)
for key, cnt in shapes.items():
    H, W, Cin, Cout, Cout_pad, k, s, pad, f32, in_ld, out_ld = key
    if ONLY and [int(v) for v in ONLY.split(",")] != [H, Cin, Cout, k, s]: continue
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    K = k * k * Cin; K_pad = (K + 63) // 64 * 64
    x = torch.randn((B, H, W, in_ld), device="cuda").half()
    w = (torch.randn((Cout_pad, K_pad), device="cuda") / K ** 0.5).half()
    b = torch.randn(Cout_pad, device="cuda")
    out = torch.empty((B, Ho, Wo, out_ld), device="cuda", dtype=torch.float32 if f32 else torch.float16)
    # `act` bits of y7t_conv2d_nhwc_f16 = the weight packing the plan would give this layer (the weights are random: only the kernel choice matters here)
    if graph.ws_s2_eligible(H, W, Cin, Cout, k, s, pad, out_ld, 0, f32, in_ld, 0, B): code = ACT | 65536      # the 64 -> 128 stride-2 layer, weights stationary (Y7T_CONV_WS_S2=0: off)
    elif graph.ws128_eligible(H, W, Cin, Cout, k, s, pad, out_ld, 0, f32, in_ld, 0, B) or graph.ws128_s2_eligible(H, W, Cin, Cout, k, s, pad, out_ld, 0, f32, in_ld, 0, B): code = ACT | 16384        # the 128 -> 128 k layers, weights stationary (Y7T_CONV_WS128; this entry point has no tile counter: static partition)
    elif graph.ws_eligible(H, W, Cin, Cout, k, s, pad, out_ld, 0, f32, in_ld, 0, B): code = ACT | 8192               # opt-in: Y7T_CONV_WS=1 (weights stationary in registers)
    elif graph.patch_eligible(H, W, Cin, Cout, k, s, pad, out_ld, 0, f32, B) and KORDER: code = ACT | 1024
    elif graph.patch_s2_eligible(Cin, Cout, k, s, pad, out_ld, 0, f32, B * Ho * Wo): code = ACT | 4096          # stride-2 patch kernel where it measured faster
    elif k == 3 and Cin % 64 == 0: code = ACT | (KORDER << 8)
    elif graph.p8_eligible(H, W, Cin, Cout, k, s, out_ld, 0, f32, in_ld, 0, B): code = ACT | 32768            # 1x1, Cout % 256 == 0: the 256 x 256 x 64 ping-pong pipeline (Y7T_CONV_P8=0: off)
    elif k == 1 and Cin % 32 == 0 and KORDER and os.environ.get('Y7T_CONV_WPANEL', '1') != '0': code = ACT | 2048
    else: code = 1
    def run():
        _lib.check(L.y7t_conv2d_nhwc_f16(_lib.ptr(x), in_ld, 0, B, H, W, Cin, _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), out_ld, 0, f32, Cout, Cout_pad,
                                         k, k, s, pad, code, _lib.ptr(zeros), _lib.stream_ptr()))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2.0 * B * Ho * Wo * Cout * k * k * Cin
    byts = B * (H * W * Cin * 2 + Ho * Wo * Cout * (4 if f32 else 2)) + Cout * K * 2
    bn = 128 if Cout_pad % 128 == 0 else 64
    blocks = ((B * Ho * Wo + 127) // 128) * (Cout_pad // bn)
    tot_t += us * cnt; tot_f += flops * cnt
    print("%4dx%-4d %4d->%-4d %d/%d ld%d->%d %3d %9.1f %8.1f %8.0f %8d" % (H, W, Cin, Cout, k, s, in_ld, out_ld, cnt, us, flops / us / 1e6, byts / us / 1e3, blocks))
print("TOTAL %.3f ms per %d frames  ->  %.1f TFLOP/s" % (tot_t / 1e3, B, tot_f / tot_t / 1e6))
