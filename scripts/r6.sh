#!/bin/bash
# Round 6's GPU script (same conventions as scripts/gpu_round.sh: every step under its own `timeout`, output under gpurun_out/$TAG/, verdicts appended to
# gpurun_out/$TAG/summary.txt; a failing step does not stop the others).
#   gpurun --timeout 1500 -- 'TAG=r6a bash scripts/r6.sh lattice crowd_tests time_large suite bench'
# Steps:
#   lattice      scripts/debug_lattice.py frame by frame under a hard timeout (a component of 160 rows: the path that had never run on a device), three scenes
#   crowd_tests  the un-skipped lattice test + the natural crowds (250 / 400 objects on 640 / 480 px) on the device
#   time_large   scripts/time_large_components.py: step time of frames on each fallback (large component on a wave, literal re-solve at 160 / 320 / 500 rows + columns)
#   suite        the whole `-m gpu` suite as the driver runs it
#   bench        the driver's bench line -> bench_line.json
#   batch_sweep  frames per step 40 / 48 / 56 / 64 (convs over runs of frames past 2 GiB)
#   perlayer     per-op table of the launch list (scripts/per_layer_table.sh)
#   profile      rocprofv3 kernel stats of the bench command + PMC passes (gpu_round.sh profile)
#   owncu_ab / hipgraph2 / mfma_ceiling / detect_nst / splitk_b1 / latency / tracker_phases / tests_tracker / tests_fullsize / nms_gate / prio_ab / cu_reserve_cfg4
#                the round's small experiments (profiles/r06_small_experiments.txt says what each one measured)
#   latency_prof / nms_ab / nms_prof   the batch-1 frame under rocprofv3; rank sort + NMS alone (scripts/time_nms.py) against lib/liby7t_prev.so (a copy of an earlier
#                build: `git stash; python -m yolov7_tracker_amd.build --product-only; cp lib/liby7t.so lib/liby7t_prev.so; git stash pop; rebuild`)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r6}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
say() { echo "=== $*" | tee -a $O/summary.txt; }
tailsum() { tail -${2:-2} $1 | tee -a $O/summary.txt; }
benchline() {  # file -> one summary line
python3 - "$1" <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    l = json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
    ph = l.get("phases_ms_per_step", {})
    lm = l.get("latency_mode") or {}
    ow = l.get("other_workloads") or {}
    print("%7.0f fps  step %.2f ms  list %.2f ms  frac %.4f  chain %.2f ms | host-fed %s coupled %s | latency u8 %s f32 %s | cfg3 %s cfg4 %s"
          % (l["value"], l["ms_per_step"], l["roofline"].get("launch_list_ms", float("nan")), l["roofline"]["frac"], ph.get("tracker_chain", float("nan")),
             (l.get("fps_incl_h2d") or {}).get("value"), "%s / confident rows %s (chain %s ms)" % ((l.get("coupled") or {}).get("fps"), (l.get("coupled_confident_rows") or {}).get("fps"), (l.get("coupled_confident_rows") or {}).get("tracker_chain_ms")),
             (lm.get("u8_hwc_host") or {}).get("fps"), (lm.get("f32_chw_host") or {}).get("fps"),
             {k: (ow.get("cfg3") or {}).get(k) for k in ("fps", "tracker_chain_ms", "launch_list_ms")}, {k: (ow.get("cfg4") or {}).get(k) for k in ("fps", "tracker_chain_ms", "launch_list_ms")}))
except Exception as e:
    print("no bench line in %s: %r" % (sys.argv[1], e))
PY
}

for step in "$@"; do case $step in

lattice)
  say "lattice: the 160-row component on the device, frame by frame, hard timeout 150 s per scene"
  for sc in "bytetrack 0" "bytetrack 60" "botsort 0"; do
    n=$(echo $sc | tr ' ' '_')
    timeout 150 python scripts/debug_lattice.py $sc > $O/lattice_$n.log 2>&1; echo "rc=$?" >> $O/lattice_$n.log
    echo "--- $sc" | tee -a $O/summary.txt; grep -c "oracle's: True" $O/lattice_$n.log | sed 's/^/frames equal to the oracle: /' | tee -a $O/summary.txt; tailsum $O/lattice_$n.log 2
  done
  ;;

crowd_tests)
  say "crowd_tests: lattice + natural crowds on the device (tests/test_tracker_gpu.py)"
  timeout 600 python -m pytest -q -m gpu tests/test_tracker_gpu.py -k "larger_than_a_wave or crowded" > $O/t_crowd.log 2>&1; echo "rc=$?" >> $O/t_crowd.log; tailsum $O/t_crowd.log 4
  ;;

time_large)
  say "time_large: step time of the frames on each fallback"
  timeout 600 python scripts/time_large_components.py > $O/time_large.log 2>&1; echo "rc=$?" >> $O/time_large.log
  cat $O/time_large.log | cut -c1-260 | tee -a $O/summary.txt
  ;;

suite)
  say "suite: python -m pytest tests -x -q -m gpu"
  timeout 2400 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; echo "rc=$?" >> $O/suite.log; tailsum $O/suite.log 4
  ;;

bench)
  say "bench: python bench.py --steps 20 --warmup 5"
  timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "rc=$?" | tee -a $O/summary.txt
  benchline $O/bench_line.json
  ;;

batch_sweep)
  say "batch_sweep: frames per step above the 40 that 2 GiB tensors allowed (conv launches over runs of frames, csrc/y7t_detector.hip): the chunk test, then the headline at 40 / 48 / 56 / 64 / 40"
  timeout 600 python -m pytest -x -q -m gpu tests/test_fullsize_gpu.py -k "runs_of_frames" > $O/t_runs.log 2>&1; echo "rc=$?" >> $O/t_runs.log; tailsum $O/t_runs.log 3
  for b in ${SWEEP:-40 48 56 64 40}; do
    timeout 900 python bench.py --steps 12 --warmup 4 --batch $b --no_latency_mode --no_cpu_baseline --no_other_workloads --no_coupled > $O/bench_b$b.json 2> $O/bench_b$b.err; echo "b$b rc=$?" | tee -a $O/summary.txt
    benchline $O/bench_b$b.json; tail -2 $O/bench_b$b.err | cut -c1-300
  done
  ;;

perlayer_b)
  say "perlayer_b: per-op tables at B=${B:-80} frames: the default lowering, every eligible 1x1 on p8 (Y7T_CONV_P8=all), 64-row panels kept for the small maps (as at 40 frames)"
  E="Y7T_LIB=$ROOT/yolov7-tracker_amd/lib/liby7t_ablate.so"
  for v in ${VARIANTS:-default p8all panel64}; do
    case $v in default) X="";; p8all) X="Y7T_CONV_P8=all";; p8t200) X="Y7T_CONV_P8=all Y7T_CONV_P8_MIN_TILES=200";; s2all) X="Y7T_CONV_PATCH_S2=1";; nopersist) X="Y7T_CONV_P8_PERSIST=0";; panel64) X="Y7T_CONV_PATCH_PANEL64_BELOW=512 Y7T_CONV_1X1_PANEL64_BELOW=1000";; esac
    env $E $X B=${B:-80} NAME=b${B:-80}_$v OUT=$O timeout 600 bash scripts/per_layer_table.sh > $O/pl_b${B:-80}_$v.log 2>&1
    echo "--- $v" | tee -a $O/summary.txt; tail -1 $O/per_layer_b${B:-80}_$v.txt | tee -a $O/summary.txt
  done
  ;;

tests_pinned)
  say "tests_pinned: the pinned-configuration detector tests (80 frames) + the full-size property tests"
  timeout 2400 python -m pytest -x -q -m gpu tests/test_detector_pinned_gpu.py tests/test_fullsize_gpu.py > $O/t_pinned.log 2>&1; echo "rc=$?" >> $O/t_pinned.log; tailsum $O/t_pinned.log 4
  ;;

bench_quick)
  say "bench_quick: the headline only (no latency mode / cpu baseline / other workloads)"
  timeout 600 python bench.py --steps 20 --warmup 5 --no_latency_mode --no_cpu_baseline --no_other_workloads > $O/bench_quick.json 2> $O/bench_quick.err; echo "rc=$?" | tee -a $O/summary.txt
  benchline $O/bench_quick.json
  ;;

perlayer)
  say "perlayer: per-op table of the launch list at 40 frames"
  NAME=${NAME:-$TAG} OUT=$O timeout 600 bash scripts/per_layer_table.sh > $O/per_layer.log 2>&1; echo "rc=$?" | tee -a $O/summary.txt
  tail -3 $O/per_layer_${NAME:-$TAG}.txt | cut -c1-300 | tee -a $O/summary.txt
  ;;

profile)
  TAG=$TAG bash scripts/gpu_round.sh profile
  ;;

owncu_ab)
  say "owncu_ab: tracker step kernels asking for the CU's whole LDS (default) against asking for what they use (measuring build, Y7T_TRACKER_OWN_CU=0): headline + cfg3 / cfg4 child runs, A/B/A"
  for v in own1_a own0 own1_b; do
    case $v in own0) E="Y7T_LIB=$ROOT/yolov7-tracker_amd/lib/liby7t_ablate.so Y7T_TRACKER_OWN_CU=0";; *) E="Y7T_LIB=$ROOT/yolov7-tracker_amd/lib/liby7t_ablate.so Y7T_TRACKER_OWN_CU=1";; esac
    env $E timeout 900 python bench.py --steps 10 --warmup 3 --no_latency_mode --no_cpu_baseline --no_coupled > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?" | tee -a $O/summary.txt
    benchline $O/bench_$v.json
  done
  ;;

hipgraph2)
  say "hipgraph2: the conv launch list as two captured hipGraphs per candidate set (bench.py --hipgraph 2) against the plain launches, A/B/A"
  for v in plain_a graph plain_b; do
    case $v in graph) F="--hipgraph 2";; *) F="";; esac
    timeout 600 python bench.py --steps 20 --warmup 5 --no_latency_mode --no_cpu_baseline --no_other_workloads --no_coupled $F > $O/bench_hg_$v.json 2> $O/bench_hg_$v.err; echo "$v rc=$?" | tee -a $O/summary.txt
    benchline $O/bench_hg_$v.json
  done
  ;;

mfma_ceiling)
  say "mfma_ceiling: the register-only MFMA loop on random / zero operands (scripts/ubench/mfma_power.hip), re-measured this round"
  ( cd scripts/ubench && bash build.sh > $O/ubench_build.log 2>&1 )
  timeout 300 scripts/ubench/mfma_power > $O/mfma_power.txt 2>&1; echo "rc=$?" | tee -a $O/summary.txt
  cat $O/mfma_power.txt | cut -c1-200 | tee -a $O/summary.txt
  ;;

detect_nst)
  say "detect_nst: Detect convs on the four-stage ring (default) against the two-stage one (measuring build, Y7T_CONV_DETECT_NST=2): parity tests, per-op tables at 40 frames and at one frame"
  timeout 900 python -m pytest -x -q -m gpu tests/test_detector_pinned_gpu.py -k "fused_detect or candidates or launch_list" > $O/t_detect.log 2>&1; echo "rc=$?" >> $O/t_detect.log; tailsum $O/t_detect.log 3
  for v in nst4 nst2; do
    case $v in nst2) E="Y7T_LIB=$ROOT/yolov7-tracker_amd/lib/liby7t_ablate.so Y7T_CONV_DETECT_NST=2";; *) E="Y7T_LIB=$ROOT/yolov7-tracker_amd/lib/liby7t_ablate.so";; esac
    env $E NAME=b40_$v OUT=$O timeout 600 bash scripts/per_layer_table.sh > $O/pl_b40_$v.log 2>&1
    env $E B=1 NAME=b1_$v OUT=$O timeout 600 bash scripts/per_layer_table.sh > $O/pl_b1_$v.log 2>&1
    for b in b40 b1; do echo "--- $b $v" | tee -a $O/summary.txt; grep "detect-decode\|TOTAL" $O/per_layer_${b}_$v.txt | cut -c1-200 | tee -a $O/summary.txt; done
  done
  ;;

splitk_b1)
  say "splitk_b1: the batch-1 launch list without split-K (measuring build, Y7T_CONV_SPLITK=0) against the default, per-op tables"
  E="Y7T_LIB=$ROOT/yolov7-tracker_amd/lib/liby7t_ablate.so"
  env $E Y7T_CONV_SPLITK=0 B=1 NAME=b1_nosplit OUT=$O timeout 600 bash scripts/per_layer_table.sh > $O/pl_b1_nosplit.log 2>&1
  tail -1 $O/per_layer_b1_nosplit.txt | tee -a $O/summary.txt
  ;;

latency)
  say "latency: bench.py's latency mode alone (reference Timer semantics, batch 1), three runs"
  for k in 1 2 3; do
    timeout 600 python scripts/latency_mode.py > $O/latency_$k.log 2>&1; echo "rc=$?" >> $O/latency_$k.log; grep -h "fps" $O/latency_$k.log | cut -c1-300 | tee -a $O/summary.txt
  done
  ;;

tracker_phases)
  say "tracker_phases: scripts/time_tracker.py (frame step alone: latency, kernel-only, phase table at 80 and 500 objects)"
  timeout 600 python scripts/time_tracker.py > $O/time_tracker.log 2>&1; echo "rc=$?" >> $O/time_tracker.log; cat $O/time_tracker.log | cut -c1-400 | tee -a $O/summary.txt
  ;;

tests_tracker)
  say "tests_tracker: tracker / CLI / chained / multirank device tests"
  timeout 1500 python -m pytest -x -q -m gpu tests/test_tracker_gpu.py tests/test_cli_gpu.py tests/test_chained_gpu.py tests/test_multirank_gpu.py > $O/t_tracker.log 2>&1; echo "rc=$?" >> $O/t_tracker.log; tailsum $O/t_tracker.log 3
  ;;

nms_gate)
  say "nms_gate: where the previous batch's rank sort + NMS is released into the forward (bench.py --nms_gate_div 8 / 16 / 32), A/B/C/A"
  for v in 8 16 32 8; do
    timeout 600 python bench.py --steps 20 --warmup 5 --no_latency_mode --no_cpu_baseline --no_other_workloads --no_coupled --nms_gate_div $v > $O/bench_gate_$v.json 2> $O/bench_gate_$v.err; echo "gate_div $v rc=$?" | tee -a $O/summary.txt
    benchline $O/bench_gate_$v.json
  done
  ;;

latency_prof)
  say "latency_prof: rocprofv3 kernel stats of the batch-1 latency loop (scripts/latency_mode.py): which kernels a frame's 2.5 ms are"
  ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/lp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- python $ROOT/scripts/latency_mode.py 40 > $O/latency_prof.log 2>&1
    f=$(find /tmp/lp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/latency_kernel_stats.csv )
  head -25 $O/latency_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
  ;;

nms_ab)
  say "nms_ab: rank sort + kept-list NMS at one frame and at 40 (scripts/time_nms.py): the previous build (lib/liby7t_prev.so) / this one, A/B/A; NMS parity tests"
  for v in prev new prev2 new2; do
    case $v in prev*) E="Y7T_LIB=$ROOT/yolov7-tracker_amd/lib/liby7t_prev.so";; *) E="Y7T_X=0";; esac
    env $E timeout 300 python scripts/time_nms.py > $O/nms_$v.log 2>&1; echo "--- $v rc=$?" | tee -a $O/summary.txt; grep "^B=" $O/nms_$v.log | cut -c1-200 | tee -a $O/summary.txt
  done
  timeout 900 python -m pytest -x -q -m gpu tests/test_detector_gpu.py tests/test_detector_pinned_gpu.py -k "nms or boxes or postprocess or candidates" > $O/t_nms.log 2>&1; echo "rc=$?" >> $O/t_nms.log; tailsum $O/t_nms.log 3
  ;;

nms_prof)
  say "nms_prof: rocprofv3 kernel stats of scripts/time_nms.py (min = one frame, max = 40 frames)"
  ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/np
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np -- python $ROOT/scripts/time_nms.py > $O/nms_prof.log 2>&1
    f=$(find /tmp/np -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/nms_kernel_stats.csv )
  grep "k_rank_sort\|k_nms_keep" $O/nms_kernel_stats.csv | cut -c1-40,150-260 | tee -a $O/summary.txt
  ;;

deepsort_prof)
  say "deepsort_prof: scripts/time_deepsort.py alone (kernel-only us per frame, phase stamps), then its kernels under rocprofv3 --kernel-trace --stats"
  timeout 600 python scripts/time_deepsort.py > $O/time_deepsort.log 2>&1; echo "rc=$?" >> $O/time_deepsort.log; cut -c1-300 $O/time_deepsort.log | tee -a $O/summary.txt
  ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/dp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -- python $ROOT/scripts/time_deepsort.py > $O/deepsort_prof.log 2>&1
    f=$(find /tmp/dp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/deepsort_kernel_stats.csv )
  head -12 $O/deepsort_kernel_stats.csv | cut -c1-60,100-260 | tee -a $O/summary.txt
  ;;

cfg4_trace)
  say "cfg4_trace: rocprofv3 --kernel-trace of a short cfg4 bench (where a DeepSORT frame's four launches wait beside the detector) -> cfg4_kernel_trace.csv"
  ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/c4
    timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/c4 -- python $ROOT/bench.py --workload ${WL:-cfg4} --steps 3 --warmup 1 --no_latency_mode --no_cpu_baseline --no_other_workloads --no_coupled > $O/cfg4_trace.log 2>&1
    f=$(find /tmp/c4 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $ROOT/scripts/trace_compact.py $f $O/${WL:-cfg4}_kernel_trace.csv )
  wc -l $O/${WL:-cfg4}_kernel_trace.csv | tee -a $O/summary.txt; tail -2 $O/cfg4_trace.log | cut -c1-300
  ;;

closing_ab)
  say "closing_ab: the 80-frame headline against (b) persistent conv kernels on a static partition (Y7T_CONV_WS_DYN=0), (c) 96 frames, (d) the list as captured hipGraphs, (e) NMS gate variants; A first and last"
  Q="--steps 16 --warmup 4 --no_latency_mode --no_cpu_baseline --no_other_workloads --no_coupled"
  run() { n=$1; shift; env "$@" > /dev/null 2>&1; }
  timeout 600 python bench.py $Q > $O/ab_a.json 2> $O/ab_a.err; echo "a default rc=$?" | tee -a $O/summary.txt; benchline $O/ab_a.json
  Y7T_CONV_WS_DYN=0 timeout 600 python bench.py $Q > $O/ab_b.json 2> $O/ab_b.err; echo "b static partition rc=$?" | tee -a $O/summary.txt; benchline $O/ab_b.json
  timeout 600 python bench.py $Q --batch 96 > $O/ab_c.json 2> $O/ab_c.err; echo "c 96 frames rc=$?" | tee -a $O/summary.txt; benchline $O/ab_c.json
  timeout 600 python bench.py $Q --hipgraph 2 > $O/ab_d.json 2> $O/ab_d.err; echo "d hipgraph 2 rc=$?" | tee -a $O/summary.txt; benchline $O/ab_d.json
  timeout 600 python bench.py $Q --nms_gate_div 16 > $O/ab_e.json 2> $O/ab_e.err; echo "e nms gate 16 rc=$?" | tee -a $O/summary.txt; benchline $O/ab_e.json
  timeout 600 python bench.py $Q --tracker_launch per_frame > $O/ab_f.json 2> $O/ab_f.err; echo "f tracker per frame rc=$?" | tee -a $O/summary.txt; benchline $O/ab_f.json
  timeout 600 python bench.py $Q > $O/ab_a2.json 2> $O/ab_a2.err; echo "a2 default rc=$?" | tee -a $O/summary.txt; benchline $O/ab_a2.json
  ;;

threads_sweep)
  say "threads_sweep: threads of the tracker step workgroup inside the pipeline (bench.py --tracker_threads), per workload"
  for wl in cfg2 cfg3 cfg4; do for t in ${THREADS:-0 128 256 512}; do
    timeout 600 python bench.py --workload $wl --steps 8 --warmup 3 --no_latency_mode --no_cpu_baseline --no_other_workloads --no_coupled --tracker_threads $t > $O/bench_thr_${wl}_$t.json 2> $O/bench_thr_${wl}_$t.err; echo "$wl threads $t rc=$?" | tee -a $O/summary.txt
    benchline $O/bench_thr_${wl}_$t.json
  done; done
  ;;

cfg4_ab)
  say "cfg4_ab: DeepSORT device tests, scripts/time_deepsort.py, then the cfg4 bench (k_embed_dist over the live slots)"
  timeout 900 python -m pytest -x -q -m gpu tests/test_tracker_gpu.py tests/test_reid_gpu.py -k "deepsort or DeepSORT or reid or embed" > $O/t_ds.log 2>&1; echo "rc=$?" >> $O/t_ds.log; tailsum $O/t_ds.log 3
  timeout 600 python scripts/time_deepsort.py > $O/time_deepsort.log 2>&1; grep "kernel-only" $O/time_deepsort.log | cut -c1-200 | tee -a $O/summary.txt
  for k in 1 2; do
    timeout 600 python bench.py --workload cfg4 --steps 8 --warmup 3 --no_latency_mode --no_cpu_baseline --no_other_workloads --no_coupled > $O/bench_cfg4_$k.json 2> $O/bench_cfg4_$k.err; echo "cfg4 run $k rc=$?" | tee -a $O/summary.txt
    benchline $O/bench_cfg4_$k.json
  done
  ;;

prio_ab)
  say "prio_ab: the tracker chain's stream at high queue priority (bench.py --prio 2) against the default, cfg4 / cfg3 / cfg2, A/B/A"
  for wl in cfg4 cfg3 cfg2; do for v in 0 2 0b; do
    timeout 600 python bench.py --workload $wl --steps 8 --warmup 3 --no_latency_mode --no_cpu_baseline --no_other_workloads --no_coupled --prio ${v%b} > $O/bench_prio_${wl}_$v.json 2> $O/bench_prio_${wl}_$v.err; echo "$wl prio $v rc=$?" | tee -a $O/summary.txt
    benchline $O/bench_prio_${wl}_$v.json
  done; done
  ;;

cu_reserve_cfg4)
  say "cu_reserve_cfg4: the tracker chain's stream on reserved CUs (bench.py --cu_reserve N) for cfg4 (four dependent launches per frame) and cfg3, against the default"
  for wl in cfg4 cfg3; do for v in 0 2 8 0b; do
    timeout 600 python bench.py --workload $wl --steps 8 --warmup 3 --no_latency_mode --no_cpu_baseline --no_other_workloads --no_coupled --cu_reserve ${v%b} > $O/bench_cur_${wl}_$v.json 2> $O/bench_cur_${wl}_$v.err; echo "$wl cu_reserve $v rc=$?" | tee -a $O/summary.txt
    benchline $O/bench_cur_${wl}_$v.json
  done; done
  ;;

reduce_ab)
  say "reduce_ab: k_splitk_reduce with its slab loads issued four at a time against the previous build (lib/liby7t_prev.so): batch-1 per-op tables A/B/A, layer parity tests"
  for v in prev new prev2 new2; do
    case $v in prev*) E="Y7T_LIB=$ROOT/yolov7-tracker_amd/lib/liby7t_prev.so";; *) E="Y7T_X=0";; esac
    env $E B=1 NAME=b1_$v OUT=$O timeout 600 bash scripts/per_layer_table.sh > $O/pl_b1_$v.log 2>&1
    echo "--- $v" | tee -a $O/summary.txt; tail -1 $O/per_layer_b1_$v.txt | tee -a $O/summary.txt
  done
  timeout 900 python -m pytest -x -q -m gpu tests/test_detector_gpu.py -k "conv or split or layer" > $O/t_conv.log 2>&1; echo "rc=$?" >> $O/t_conv.log; tailsum $O/t_conv.log 3
  ;;

prev_ab)
  say "prev_ab: per-op tables at 80 frames, this build against lib/liby7t_prev.so, A/B/A/B; then the p8 / stem / pinned parity tests"
  for v in prev new prev2 new2; do
    case $v in prev*) E="Y7T_LIB=$ROOT/yolov7-tracker_amd/lib/liby7t_prev.so";; *) E="Y7T_X=0";; esac
    env $E NAME=b80_$v OUT=$O timeout 600 bash scripts/per_layer_table.sh > $O/pl_b80_$v.log 2>&1
    echo "--- $v" | tee -a $O/summary.txt; grep "${ROWS:-p8\|stem}" $O/per_layer_b80_$v.txt | awk '{u+=$(NF-5)} END {print "rows of interest:", u, "us"}' | tee -a $O/summary.txt; tail -1 $O/per_layer_b80_$v.txt | tee -a $O/summary.txt
  done
  timeout 900 python -m pytest -x -q -m gpu tests/test_detector_gpu.py -k "pingpong or stem or u8" > $O/t_p8.log 2>&1; echo "rc=$?" >> $O/t_p8.log; tailsum $O/t_p8.log 3
  timeout 900 python -m pytest -x -q -m gpu tests/test_detector_pinned_gpu.py -k "every_op or launch_list" > $O/t_pin.log 2>&1; echo "rc=$?" >> $O/t_pin.log; tailsum $O/t_pin.log 3
  ;;

tests_fullsize)
  say "tests_fullsize: BASELINE-size properties incl. cfg3 (300 frames x 500 objects, BoT-SORT) against the oracle"
  timeout 1500 python -m pytest -x -q -m gpu tests/test_fullsize_gpu.py > $O/t_fullsize.log 2>&1; echo "rc=$?" >> $O/t_fullsize.log; tailsum $O/t_fullsize.log 3
  ;;

*) say "unknown step $step";;
esac; done
