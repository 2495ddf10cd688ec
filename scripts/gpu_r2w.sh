#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r2w; mkdir -p $O
timeout 900 python -m pytest tests/test_tracker_gpu.py tests/test_fullsize_gpu.py tests/test_cli_gpu.py tests/test_multirank_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
timeout 300 python scripts/time_tracker.py 2>&1 | head -6
timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/cfg2.log 2>&1
echo "cfg2: $(grep -o '"value": [0-9.]*, "unit": "frames/s"\|"tracker_chain": [0-9.]*\|"detector_forward": [0-9.]*' $O/cfg2.log | tr '\n' ' ')"
timeout 600 python bench.py --workload cfg3 --steps 10 --warmup 3 --no_cpu_baseline > $O/cfg3.log 2>&1
echo "cfg3: $(grep -o '"value": [0-9.]*, "unit": "frames/s"\|"tracker_chain": [0-9.]*' $O/cfg3.log | tr '\n' ' ')"
