#!/bin/bash
# end-of-round consolidated run: full GPU suite, smoke, the three bench lines, profiles (run on the GPU box from the repo root)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r2u; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 2 $O/tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; echo "bench rc=$?"; tail -n 1 $O/bench_default.log > $O/bench_line.json
timeout 600 python bench.py --workload cfg3 --steps 10 --warmup 3 --no_cpu_baseline > $O/bench_cfg3.log 2>&1; tail -n 1 $O/bench_cfg3.log > $O/bench_line_cfg3.json
timeout 600 python bench.py --workload cfg4 --steps 10 --warmup 3 > $O/bench_cfg4.log 2>&1; tail -n 1 $O/bench_cfg4.log > $O/bench_line_cfg4.json
for f in bench_line bench_line_cfg3 bench_line_cfg4; do echo "$f: $(grep -o '"value": [0-9.]*, "unit": "frames/s"\|"launch_list_ms": [0-9.]*\|"frac": [0-9.]*\|"tracker_chain": [0-9.]*' $O/$f.json | head -5 | tr '\n' ' ')"; done
timeout 300 python scripts/time_deepsort.py > $O/time_deepsort.txt 2>&1
timeout 300 python scripts/time_reid.py > $O/time_reid.txt 2>&1
timeout 300 python scripts/time_tracker.py > $O/time_tracker.txt 2>&1
bash scripts/profile_r02.sh > $O/profile.log 2>&1; echo "profile rc=$?"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4 -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 5 --warmup 2 --no_cpu_baseline > $O/cfg4_under_rocprof.log 2>&1
f=$(find /tmp/k4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg4_kernel_stats.csv
ls $GRAFT_REPO_ROOT/gpurun_out/prof2 | head
