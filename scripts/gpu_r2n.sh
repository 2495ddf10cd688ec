#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_tracker_gpu.py tests/test_fullsize_gpu.py tests/test_cli_gpu.py -x -q -m gpu ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log
timeout 300 python scripts/time_tracker.py > $O/time_tracker.log 2>&1; grep -v amdgpu.ids $O/time_tracker.log | cut -c1-330
( timeout 600 python bench.py --workload cfg3 --no_cpu_baseline ) > $O/bench_cfg3.log 2>&1; echo "cfg3 rc=$?"
( timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode ) > $O/bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("bench_cfg3","bench"):
    for l in open("gpurun_out/r2n/%s.log"%f):
        if l.startswith("{"):
            d=json.loads(l); print(f, {k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["phases_ms_per_step"]["detector_forward"], d["phases_ms_per_step"]["decode_nms"], d["phases_ms_per_step"]["tracker_chain"])
PY
