#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2h; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_detector_gpu.py tests/test_detector_pinned_gpu.py tests/test_cli_gpu.py -x -q -m gpu ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
cd /tmp; rm -rf /tmp/prof
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_latency_mode ) > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find /tmp/prof -name "*kernel_trace.csv"); do cp $f $O/kernel_trace.csv; done
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
ls $O
