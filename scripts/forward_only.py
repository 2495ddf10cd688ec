"""N forwards of the benchmarked detector configuration and nothing else (for rocprofv3 passes whose per-dispatch rows must be attributed
to ops): YOLOv7-w6 @ 1280, bench.DEFAULT_BATCH uint8 frames resident in HBM, fused stem + fused Detect decode -- the launch list bench.py times.
Prints the launch list (op index -> kernel variant, shape, algorithmic GFLOP / bytes) as JSON on the last line."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov7_tracker_amd import synth
from yolov7_tracker_amd.detector import arch, model

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
import hashlib, types
import bench       # the timed run's weights (bench.py --weights conditioned, its default) and its frames per step
B = int(os.environ.get("B", str(bench.DEFAULT_BATCH)))
WEIGHTS = os.environ.get("WEIGHTS", "conditioned")
frames_host = synth.make_frames(B, 80, 1280, seq_idx=0)
sd = bench.conditioned_state_dict(types.SimpleNamespace(arch="yolov7-w6", img=1280), 10, frames_host) if WEIGHTS == "conditioned" else None
det = model.Detector(arch.yolov7_w6(10), sd, img_size=(1280, 1280), max_batch=B, seed=0)
frames = torch.from_numpy(frames_host).cuda()
bench.plant_objectness_bias(det, frames)      # all four Detect levels live, each supplying its quota of the candidates, as bench.py
det.forward(frames)
names = det.launch_list(B)
LL_SHA = hashlib.sha1(json.dumps(names).encode()).hexdigest()[:16]      # == bench.py's config.launch_list_sha for the same list
torch.cuda.synchronize()
print("MARK forwards begin")
for i in range(N):
    det.forward(frames, fuse_decode=0.01, pset=i % 2)
    torch.cuda.synchronize()
p = det.plan
ops, ci = [], 0
n_launches = det.launches_per_op(B)      # (a conv whose tensors pass 2 GiB at B frames: several launches over runs of frames)
for oi, op in enumerate(p.ops):
    if int(op["type"]) == 0 and int(op["detect_level"]) >= 0 and p.fusable:
        nst = 4 if (B * int(op["Ho"]) * int(op["Wo"]) + 127) // 128 <= 4096 and os.environ.get("Y7T_CONV_DETECT_NST", "4") == "4" else 2      # (csrc/y7t_conv.hip::conv_dispatch, the Detect branch)
        names[oi] = "igemm<128,64,32,%d> 1x1 detect-decode" % nst      # (launch_list probes the ops in plain mode; the fused forward never splits K here)
    d = {"op": oi, "kernel": names[oi], "H": int(op["H"]), "W": int(op["W"]), "Cin": int(op["Cin"]), "Cout": int(op["Cout"]), "k": int(op["KH"]), "s": int(op["stride"]), "launches": n_launches[oi]}
    if int(op["type"]) == 0:
        wl = p.wlayout[ci]; ci += 1
        macs = wl["macs"]
        if wl.get("fused_next"):      # korder 11: the stride-2 layer + the twin 1x1 behind it are ONE launch (two weight banks)
            macs += p.wlayout[ci]["macs"]; ci += 1
        d["gflop"] = 2 * macs * B / 1e9
        cin_bytes = (int(op["Cin"]) - int(op["up_C"])) * int(op["H"]) * int(op["W"]) + int(op["up_C"]) * int(op["H"]) * int(op["W"]) // 4
        if oi == 0 and p.stem_fused:
            cin_bytes = 3 * 1280 * 1280 // 2          # uint8 frame (bytes), expressed in fp16 elements
        out_elems = int(op["Ho"]) * int(op["Wo"]) * int(op["Cout"]) if int(op["detect_level"]) < 0 else 0
        d["bytes"] = (cin_bytes + out_elems) * 2 * B + int(op["Cout_pad"]) * int(op["K_pad"]) * 2
    ops.append(d)
print(json.dumps({"B": B, "forwards": N, "weights": WEIGHTS, "launch_list_sha": LL_SHA, "ops": ops}))
