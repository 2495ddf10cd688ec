#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_detector_gpu.py tests/test_fullsize_gpu.py tests/test_detector_pinned_gpu.py tests/test_cli_gpu.py tests/test_tracker_gpu.py -x -q -m gpu ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log
( timeout 600 python bench.py ) > $O/bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r2m/bench.log"):
    if l.startswith("{"):
        d=json.loads(l); print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["phases_ms_per_step"]["detector_forward"], d["phases_ms_per_step"]["decode_nms"], d["phases_ms_per_step"]["tracker_chain"], d.get("fps_incl_h2d",{}).get("value"), d.get("latency_mode"), d.get("parity"))
PY
