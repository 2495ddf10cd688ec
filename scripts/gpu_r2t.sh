#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r2t; mkdir -p $O
timeout 600 python -m pytest tests/test_reid_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
Y7T_REID_PROF=1 timeout 300 python scripts/time_reid.py 2>&1 | grep -m2 "osnet"
timeout 300 python scripts/time_reid.py 2>&1 | grep crops
