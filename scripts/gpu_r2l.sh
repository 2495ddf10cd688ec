#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log
( timeout 600 python bench.py ) > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-400
( timeout 600 python bench.py --workload cfg3 --no_cpu_baseline ) > $O/bench_cfg3.log 2>&1; echo "cfg3 rc=$?"; tail -1 $O/bench_cfg3.log | cut -c1-1500
bash scripts/profile_r02.sh > $O/profile.log 2>&1; tail -12 $O/profile.log | cut -c1-400
