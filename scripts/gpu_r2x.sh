#!/bin/bash
# round 2: the reference's DeepSORT embedding network on the device op list + the OSNet op-list tests that share its kernels
mkdir -p gpurun_out/r2x
timeout 200 python -m pytest tests/test_reid_gpu.py -x -q -m gpu > gpurun_out/r2x/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2x/tests.log
tail -15 gpurun_out/r2x/tests.log
timeout 100 python - > gpurun_out/r2x/time_deepsort_net.txt 2>&1 <<'PY'
import time, torch
from yolov7_tracker_amd.tracker import reid
e = reid.ReIDExtractor(None, arch="deepsort", max_crops=80)
x = torch.randn((80, 128, 64, 3), device="cuda")
for _ in range(2): e.forward_crops(x)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3): e.forward_crops(x)
torch.cuda.synchronize(); dt = (time.time() - t0) / 3
print("deepsort_reid.Net, fp32 op list: 80 crops %.1f ms (%.2f TFLOP/s of 2.2 GFLOP per crop)" % (dt * 1e3, 80 * 2.2e9 / dt / 1e12))
PY
cat gpurun_out/r2x/time_deepsort_net.txt | tail -3
