#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_detector_gpu.py tests/test_detector_pinned_gpu.py tests/test_cli_gpu.py tests/test_tracker_gpu.py -x -q -m gpu ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
for fused in 1 0; do
( Y7T_STEM_FUSED=$fused timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_latency_mode ) > $O/bench_stem$fused.log 2>&1
echo "stem fused $fused:"; grep -o '"value": [0-9.]*, "unit": "frames/s"\|"launch_list_ms": [0-9.]*\|"decode_nms": [0-9.]*' $O/bench_stem$fused.log | head -4
done
