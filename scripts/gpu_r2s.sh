#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r2s; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tracker_gpu.py -x -q -m gpu -k "deepsort" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
for v in "--prio 2" "--prio 0"; do
timeout 600 python bench.py --workload cfg4 --steps 10 --warmup 3 --no_cpu_baseline $v > $O/cfg4.log 2>&1
echo "cfg4 $v: $(grep -o '"value": [0-9.]*, "unit": "frames/s"\|"tracker_chain": [0-9.]*\|"reid": [0-9.]*\|"detector_forward": [0-9.]*' $O/cfg4.log | tr '\n' ' ')"
done
