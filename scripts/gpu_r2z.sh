#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r2z; mkdir -p $O
for v in "--tracker_threads 0" "--tracker_threads 512" "--tracker_threads 64" "--workload cfg4" "--workload cfg4 --tracker_threads 64"; do
  timeout 400 python bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_latency_mode $v > "$O/t.log" 2>&1
  echo "$v: $(grep -o '"value": [0-9.]*, "unit": "frames/s"\|"tracker_chain": [0-9.]*\|"detector_forward": [0-9.]*' "$O/t.log" | tr '\n' ' ')"
done
timeout 300 python -m pytest tests/test_tracker_gpu.py -x -q -m gpu 2>&1 | tail -1
