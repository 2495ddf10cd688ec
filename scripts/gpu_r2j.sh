#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
python scripts/latency_probe.py 30 > $O/latency.log 2>&1; tail -3 $O/latency.log | cut -c1-1500
cd /tmp; rm -rf /tmp/prof
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o lat -- python $GRAFT_REPO_ROOT/scripts/latency_probe.py 20 ) > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find /tmp/prof -name "*kernel_trace.csv"); do cp $f $O/kernel_trace.csv; done
