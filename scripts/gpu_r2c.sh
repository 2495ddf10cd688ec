#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_detector_gpu.py -x -q -m gpu -k "fused or upsample or head_output" ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_latency_mode ) > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find /tmp/prof -name "*.csv" | head
for f in $(find /tmp/prof -name "*kernel_trace.csv"); do cp $f $O/kernel_trace.csv; done
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
ls -la $O
