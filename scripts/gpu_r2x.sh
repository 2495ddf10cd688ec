#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r2x; mkdir -p $O
timeout 40 python scripts/dbg_botsort.py botsort_gmc 2>&1 | tail -3
timeout 300 python -m pytest tests/test_tracker_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
timeout 200 python scripts/time_tracker.py 2>&1 | grep "threads=256"
