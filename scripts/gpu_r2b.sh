#!/bin/bash
# round-2 GPU session B: fused Detect decode + upsample-on-read
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_detector_gpu.py tests/test_detector_pinned_gpu.py tests/test_fullsize_gpu.py tests/test_cli_gpu.py -x -q -m gpu ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
( time timeout 600 python bench.py --steps 10 --warmup 3 ) > $O/bench.log 2>&1
echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r2b/bench.log"):
    if l.startswith("{"):
        d=json.loads(l); print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["launch_list_ms"], d["phases_ms_per_step"]["decode_nms"], d.get("fps_incl_h2d",{}).get("value"), d.get("latency_mode"), d.get("parity"))
PY
( time timeout 600 python bench.py --steps 10 --warmup 3 --hipgraph 2 --no_cpu_baseline --no_latency_mode ) > $O/bench_graph2.log 2>&1
grep -o '"value": [0-9.]*, "unit": "frames/s"\|"launch_list_ms": [0-9.]*' $O/bench_graph2.log | head -3
( Y7T_UPSAMPLE_ON_READ=0 timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_latency_mode ) > $O/bench_noupread.log 2>&1
grep -o '"value": [0-9.]*, "unit": "frames/s"\|"launch_list_ms": [0-9.]*' $O/bench_noupread.log | head -3
