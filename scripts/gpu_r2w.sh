#!/bin/bash
mkdir -p gpurun_out/r2w
timeout 200 python -m pytest tests/test_fullsize_gpu.py tests/test_cli_gpu.py tests/test_reid_gpu.py tests/test_multirank_gpu.py -x -q -m gpu -k "stream_invariants or cli or deepsort or rank" > gpurun_out/r2w/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2w/tests.log
tail -3 gpurun_out/r2w/tests.log
