#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r2s; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tracker_gpu.py tests/test_reid_gpu.py -x -q -m gpu -k "deepsort or reid" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 2 $O/tests.log
timeout 300 python scripts/time_deepsort.py 2>&1 | tail -2 | cut -c1-330
timeout 600 python bench.py --workload cfg4 --steps 10 --warmup 3 --no_cpu_baseline > $O/cfg4.log 2>&1
echo "cfg4: $(grep -o '"value": [0-9.]*, "unit": "frames/s"\|"tracker_chain": [0-9.]*\|"reid": [0-9.]*\|"detector_forward": [0-9.]*' $O/cfg4.log | tr '\n' ' ')"
