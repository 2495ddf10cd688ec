O=gpurun_out/r5u; mkdir -p $O
timeout 900 python -m pytest tests/test_tracker_gpu.py tests/test_fullsize_gpu.py tests/test_reid_gpu.py tests/test_cli_gpu.py tests/test_multirank_gpu.py -q -m gpu -k "not w6_1280 and not conv_layer and not nms_output" 2>&1 | tail -3 | tee $O/summary.txt
for l in liby7t_prev.so liby7t.so; do Y7T_LIB=$PWD/yolov7-tracker_amd/lib/$l timeout 200 python scripts/time_tracker.py 2>&1 | grep "threads=1024\|total\|sparse assoc\|large comp"; done | tee -a $O/summary.txt
X="--steps 10 --warmup 3 --no_cpu_baseline --no_latency_mode --no_other_workloads"
for w in cfg3 cfg2; do for v in prev new prevb newb; do l=liby7t.so; [ ${v:0:4} = prev ] && l=liby7t_prev.so; Y7T_LIB=$PWD/yolov7-tracker_amd/lib/$l timeout 200 python bench.py $X --workload $w > $O/${w}_$v.json 2> $O/${w}_$v.err; python - $O/${w}_$v.json $w $v <<PY
import json,sys
l=json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
print(sys.argv[2], sys.argv[3], l["value"], "fps chain", l["phases_ms_per_step"]["tracker_chain"], "list", l["roofline"]["launch_list_ms"])
PY
done; done | tee -a $O/summary.txt
