#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r2u; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_tracker_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/tests2.log 2>&1; echo "tests rc=$?"; tail -n 2 $O/tests2.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; echo "bench rc=$?"; tail -n 1 $O/bench_default.log > $O/bench_line.json
timeout 600 python bench.py --workload cfg3 --steps 10 --warmup 3 --no_cpu_baseline > $O/bench_cfg3.log 2>&1; tail -n 1 $O/bench_cfg3.log > $O/bench_line_cfg3.json
timeout 600 python bench.py --workload cfg4 --steps 10 --warmup 3 > $O/bench_cfg4.log 2>&1; tail -n 1 $O/bench_cfg4.log > $O/bench_line_cfg4.json
for f in bench_line bench_line_cfg3 bench_line_cfg4; do echo "$f: $(grep -o '"value": [0-9.]*, "unit": "frames/s"\|"launch_list_ms": [0-9.]*\|"frac": [0-9.]*\|"tracker_chain": [0-9.]*' $O/$f.json | head -5 | tr '\n' ' ')"; done
timeout 300 python scripts/time_deepsort.py > $O/time_deepsort.txt 2>&1
timeout 300 python scripts/time_tracker.py > $O/time_tracker.txt 2>&1
