#!/bin/bash
# round-2 GPU session A: new parity tests + bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_detector_pinned_gpu.py -x -q -s -m gpu ) > gpurun_out/r2a/pinned.log 2>&1
echo "pinned rc=$?"
( time timeout 600 python -m pytest tests/test_cli_gpu.py tests/test_tracker_gpu.py "tests/test_detector_gpu.py::test_decode_nms_matches_oracle" -x -q -s -m gpu ) > gpurun_out/r2a/other.log 2>&1
echo "other rc=$?"
( time timeout 600 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r2a/bench.log 2>&1
echo "bench rc=$?"
tail -5 gpurun_out/r2a/pinned.log
tail -5 gpurun_out/r2a/other.log
tail -3 gpurun_out/r2a/bench.log
