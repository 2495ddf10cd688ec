#!/bin/bash
# round 2, late check: tracker GPU tests with the duplicate-cost tie watch + step timings
mkdir -p gpurun_out/r2v
timeout 280 python -m pytest tests/test_tracker_gpu.py -x -q -m gpu > gpurun_out/r2v/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2v/tests.log
tail -3 gpurun_out/r2v/tests.log
timeout 60 python scripts/time_tracker.py > gpurun_out/r2v/time_tracker.txt 2>&1; tail -12 gpurun_out/r2v/time_tracker.txt
timeout 60 python scripts/time_deepsort.py > gpurun_out/r2v/time_deepsort.txt 2>&1; tail -8 gpurun_out/r2v/time_deepsort.txt
