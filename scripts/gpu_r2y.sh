#!/bin/bash
# round 2: per-detector split-K workspace -- detector tests (layer level, whole network, two streams) + smoke
mkdir -p gpurun_out/r2y
timeout 240 python -m pytest tests/test_detector_gpu.py -x -q -m gpu > gpurun_out/r2y/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2y/tests.log
tail -12 gpurun_out/r2y/tests.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r2y/smoke.log
