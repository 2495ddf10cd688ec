#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2o; mkdir -p $O
export TMPDIR=/tmp
for v in "--hipgraph 0" "--hipgraph 2" "--hipgraph 1" "--halves 2"; do
  ( timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode $v ) > "$O/bench_${v// /_}.log" 2>&1
  echo "$v: $(grep -o '"value": [0-9.]*, "unit": "frames/s"\|"launch_list_ms": [0-9.]*' "$O/bench_${v// /_}.log" | tr '\n' ' ')"
done
python __graft_entry__.py smoke 2>&1 | tail -2
