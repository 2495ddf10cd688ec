#!/bin/bash
# round 2: the DeepSORT embedding network on the MFMA conv kernels; tracker tests with the clamped side kernels; timings
mkdir -p gpurun_out/r2z
timeout 150 python -m pytest tests/test_reid_gpu.py -x -q -m gpu -s -k "deepsort_embedding or osnet_forward" > gpurun_out/r2z/tests_reid.log 2>&1; echo "reid rc=$?" | tee -a gpurun_out/r2z/tests_reid.log
grep -n "max err\|passed\|failed\|Error\|error" gpurun_out/r2z/tests_reid.log | tail -12
timeout 100 python scripts/time_deepsort_net.py --fp32 gpurun_out/r2z/feats_generic.pt > gpurun_out/r2z/time_deepsort_net.txt 2>&1
Y7T_REID_PATCH=1 timeout 100 python scripts/time_deepsort_net.py gpurun_out/r2z/feats_patch.pt >> gpurun_out/r2z/time_deepsort_net.txt 2>&1
python - >> gpurun_out/r2z/time_deepsort_net.txt 2>&1 <<'PY'
import torch
a, b = torch.load("gpurun_out/r2z/feats_generic.pt"), torch.load("gpurun_out/r2z/feats_patch.pt")
print("patch vs generic kernels on 2560 crops: max |diff| %.2e of max |feature| %.3f, equal %s" % (float((a - b).abs().max()), float(a.abs().max()), bool(torch.equal(a, b))))
PY
rm -f gpurun_out/r2z/feats_*.pt
grep -v amdgpu.ids gpurun_out/r2z/time_deepsort_net.txt | tail -8
timeout 120 python -m pytest tests/test_tracker_gpu.py -x -q -m gpu -k "deepsort" > gpurun_out/r2z/tests_trk.log 2>&1; echo "tracker rc=$?" | tee -a gpurun_out/r2z/tests_trk.log
tail -2 gpurun_out/r2z/tests_trk.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r2z/smoke.log
