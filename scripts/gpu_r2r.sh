#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2r; mkdir -p $O
timeout 900 python -m pytest tests/test_reid_gpu.py tests/test_tracker_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 5 $O/tests.log
