#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_tracker_gpu.py tests/test_detector_gpu.py -x -q -m gpu ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
