#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_detector_gpu.py -x -q -m gpu -k "fused or upsample or head_output or decode" ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log
for prio in 1 0; do
( timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_latency_mode --prio $prio ) > $O/bench_prio$prio.log 2>&1
echo "prio $prio:"; grep -o '"value": [0-9.]*, "unit": "frames/s"\|"launch_list_ms": [0-9.]*\|"decode_nms": [0-9.]*' $O/bench_prio$prio.log | head -4
done
