#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r2z; mkdir -p $O
for v in "--workload cfg2 --seqs 1" "--workload cfg4 --seqs 4" "--workload cfg3 --seqs 4" "--workload cfg4 --seqs 2" "--workload cfg2 --seqs 4"; do
  timeout 400 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_latency_mode $v > "$O/b_${v// /_}.log" 2>&1
  echo "$v: $(grep -o '"value": [0-9.]*, "unit": "frames/s"\|"tracker_chain": [0-9.]*\|"detector_forward": [0-9.]*\|"tracks_alive_last_frame": [0-9]*' "$O/b_${v// /_}.log" | tr '\n' ' ') $(grep -c Traceback "$O/b_${v// /_}.log")"
done
