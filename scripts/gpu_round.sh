#!/bin/bash
# ONE parametrised GPU script (replaces the per-call scripts of rounds 1-2).  Run on the GPU box from the repo root:
#
#   gpurun --timeout 1500 -- 'TAG=r3b bash scripts/gpu_round.sh tests exp_ws bench_variants'
#   gpurun --timeout 1500 -- 'TAG=r03 bash scripts/gpu_round.sh suite bench profile'
#
# Every step runs under its own `timeout`, writes to gpurun_out/$TAG/ and appends its verdict to gpurun_out/$TAG/summary.txt; a failing step does not
# stop the others.  BEFORE calling (CPU): python -m yolov7_tracker_amd.build && git rev-parse HEAD > .commit_stamp
# Steps:
#   (round 5) r5_chained    the chained detect -> NMS -> ByteTrack parity run (tests/test_chained_gpu.py)
#             r5_wgs        persistent kernels over-decomposed (no gain)         r5_dyn       tile counter in ws64 / ws128: parity, bench A/B, per-op tables
#             r5_batch      frames per step (BATCHES="32 24 40 ...")             r5_latency   latency mode with a chunked H2D copy (measured, removed: needs --upload_pieces)
#             r5_b40        the pinned parity tests at 40 frames                 r5_branches  the Detect branches on four streams (measured, removed) + panel rules at 40 frames
#             r5_panel240   128-row panels for the 20 x 20 512-channel layers    r5_ws128s2   the stride-2 form of ws128: parity, bench A/B/A/B, per-op tables
#             suite_all     the whole `-m gpu` suite without -x                  r5_coop / r5_coop2 / r5_tracker_ab   sparse association: scalar-register row contexts, register-resident wave solve of the components, threshold variants; the tracker against a previous build (profiles/r05_tracker_association.txt)
#   tests           the GPU tests added this round (training graph, candidate parity, cfg3 full size, ADVICE cases, self-launched ranks)
#   tests4          round 4's parity tests: all four Detect levels live, full 8a bar, explained kept-set differences
#   exp_p8          the 256 x 256 x 64 ping-pong 1x1 kernel: parity, per-layer A/B against igemm<128,128,32,2>, bench lines with / without (Y7T_CONV_P8=0); exp_p8abl: its timing ablations (Y7T_CONV_ABLATE)
#   exp_lowering    round 4's small-map lowering rules against round 3's thresholds (per-op tables, bench lines with latency mode)
#   exp_r4misc      round 4's small experiments: stem with full-line stores, LDS max-pool, 64-channel tiles instead of split-K -- parity + per-op tables per variant
#   suite           the whole `-m gpu` suite, as the driver runs it
#   bench           the driver's bench line (python bench.py --steps 20 --warmup 5) -> bench_line.json
#   bench_variants  default vs --weights chaotic vs --cu_reserve 8 / 16 / 8+nms in ONE session (A/B deltas are only meaningful inside a session)
#   profile         rocprofv3 kernel stats of the bench command, per-op table of the launch list, HBM traffic (2 PMC passes), MFMA busy -> stamped JSONs
#   exp_ws          weights-stationary 64 -> 64 kernel (default; Y7T_CONV_WS=0 = the patch kernels): layer parity, parity inside the pinned list, per-layer timing, bench line
#   (round 3's first call also had exp_s2 / exp_nw8 / exp_late / exp_fixup / exp_next -- the kernels prepared at the end of round 2; their results are in
#    profiles/r03_conv_variants.txt and the losing variants are no longer in the source)
#   exp_ws4         ws64 with the instruction order spelled out: parity, per-layer time, issue counters, bench line
#   exp_issue       scripts/ubench/issue_model: what a wave issues in the shadow of its own MFMAs
#   exp_noslp       the library against lib/liby7t_prev.so (a copy of an earlier build, Y7T_LIB): parity suite, per-layer table of both, bench lines of both
#   exp_clock       shader clock per kernel (scripts/clock_probe.sh), ws64 counters at 32 frames, MFMA operand-order micro-benchmark
#   exp_wsprobe     the ws64 layer by data / layout / working set (scripts/ws_probe.py);   exp_wsabl: its timing ablations (Y7T_WS_ABLATE)
#   exp_latency     one-frame per-layer table + kernel trace of the batch-1 latency mode (scripts/latency_trace.sh)
#   exp_arena       tracker index lists in LDS for the length of a frames launch: tracker tests, bench lines with / without (Y7T_TRACKER_ARENA=0)
#   pmc_queues      which queue of the buffer->LDS path the generic kernel waits in (TA / TCP / TCC / SQ counters on three layers)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r03}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
LIBD=$ROOT/yolov7-tracker_amd/lib
cd $ROOT
say() { echo "=== $*" | tee -a $O/summary.txt; }
tailsum() { tail -${2:-2} $1 | tee -a $O/summary.txt; }
benchsum() {   # name ...: one line per bench JSON
python3 - "$O" "$@" <<'PY' | tee -a $O/summary.txt
import json, sys
o = sys.argv[1]
for n in sys.argv[2:]:
    try:
        l = json.loads([x for x in open("%s/bench_%s.json" % (o, n)).read().splitlines() if x.startswith("{")][-1])
        p, ph = l.get("parity") or {}, l.get("phases_ms_per_step", {})
        c = p.get("candidates_before_nms", {})
        print("%-14s %7.0f fps  step %.2f ms  list %.2f ms  frac %.4f  tracker chain %.2f ms  nms %.2f ms | latency %s | cands within bar %s of %s, boxes %s/%s"
              % (n, l["value"], l["ms_per_step"], l["roofline"].get("launch_list_ms", float("nan")), l["roofline"]["frac"], ph.get("tracker_chain", float("nan")),
                 ph.get("decode_nms", float("nan")), (l.get("latency_mode") or {}).get("u8_hwc_host", {}).get("fps"), c.get("frac_within_bar"), c.get("n_both"),
                 p.get("boxes_matched_same_class_1px_or_iou99_conf5e-3", p.get("boxes_matched_same_class_1px_conf5e-3")), p.get("boxes_oracle")))
        if c.get("candidates_per_detect_level"):
            print("%-14s candidates per Detect level %s, out of the coordinate bar %s; boxes by anchor row %s" % ("", c.get("candidates_per_detect_level"), c.get("n_out_of_coord_bar"), p.get("boxes_by_anchor_row")))
    except Exception as e:
        print("%-14s no bench line: %r" % (n, e))
PY
}

for step in "$@"; do case $step in

tests)
  say "tests: new GPU tests of the round"
  timeout 900 python -m pytest -x -q -m gpu -s tests/test_detector_pinned_gpu.py -k "candidates or training" > $O/t_new_detector.log 2>&1; echo "rc=$?" >> $O/t_new_detector.log
  grep -h "candidates:\|training graph, level" $O/t_new_detector.log | cut -c1-400 | tee -a $O/summary.txt; tailsum $O/t_new_detector.log
  timeout 600 python -m pytest -x -q -m gpu tests/test_fullsize_gpu.py -k cfg3 > $O/t_new_cfg3.log 2>&1; echo "rc=$?" >> $O/t_new_cfg3.log; tailsum $O/t_new_cfg3.log
  timeout 600 python -m pytest -q -m gpu tests/test_tracker_gpu.py -k "deepsort_empty or refuses or max_det" > $O/t_new_advice.log 2>&1; echo "rc=$?" >> $O/t_new_advice.log; tailsum $O/t_new_advice.log
  timeout 900 python -m pytest -q -m gpu tests/test_multirank_gpu.py > $O/t_new_multirank.log 2>&1; echo "rc=$?" >> $O/t_new_multirank.log; tailsum $O/t_new_multirank.log
  ;;

tests4)
  say "tests4: round 4 -- all four Detect levels live (damped and undamped), SURVEY 8a's full bar, every kept-set difference explained"
  timeout 900 python -m pytest -x -q -m gpu -s tests/test_detector_pinned_gpu.py -k "boxes or candidates or all_levels or heads_end_to_end_against" > $O/t_r4_parity.log 2>&1; echo "rc=$?" >> $O/t_r4_parity.log
  grep -h "candidates\|oracle keeps\|level [0-9]:" $O/t_r4_parity.log | cut -c1-700 | tee -a $O/summary.txt; tailsum $O/t_r4_parity.log 4
  ;;

exp_p8)
  say "exp_p8 a: the 256 x 256 x 64 ping-pong 1x1 kernel (csrc/y7t_conv_p8.hip): layer parity vs torch fp32, determinism under load, the DUAL loader inside a network, then inside the benchmarked list"
  timeout 400 python -m pytest tests/test_detector_gpu.py -q -m gpu -k pingpong > $O/t_p8.log 2>&1; echo "rc=$?" >> $O/t_p8.log; tailsum $O/t_p8.log 3
  timeout 500 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -k "every_op or launch_list" > $O/t_p8_pinned.log 2>&1; echo "rc=$?" >> $O/t_p8_pinned.log; tailsum $O/t_p8_pinned.log 3
  say "exp_p8 b: per-layer timing at 32 frames, 1x1 rows: igemm<128,128,32,2> (Y7T_CONV_P8=0) vs p8"
  Y7T_CONV_P8=0 timeout 200 python scripts/bench_conv.py 32 > $O/b_nop8.txt 2>&1
  timeout 200 python scripts/bench_conv.py 32 > $O/b_p8.txt 2>&1
  paste <(grep " 1/1 \|TOTAL" $O/b_nop8.txt | cut -c1-62) <(grep " 1/1 \|TOTAL" $O/b_p8.txt | cut -c36-62) | tee -a $O/summary.txt
  say "exp_p8 c: bench lines (same session): p8, then Y7T_CONV_P8=0, then p8 again"
  X="--steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode"
  timeout 300 python bench.py $X > $O/bench_p8.json 2> $O/bench_p8.err
  Y7T_CONV_P8=0 timeout 300 python bench.py $X > $O/bench_nop8.json 2> $O/bench_nop8.err
  timeout 300 python bench.py $X > $O/bench_p8b.json 2> $O/bench_p8b.err
  benchsum p8 nop8 p8b
  ;;

exp_p8abl)
  say "exp_p8abl: timing ablations of the p8 kernel (wrong results): Y7T_CONV_ABLATE 0 full, 1 no in-loop DMAs, 2 no MFMAs, 4 no fragment reads, 6 neither, 7 barriers only, 8 no setprio, 16 no stores"
  for shape in ${P8SHAPES:-80,1024,512,1,1 160,512,256,1,1 40,1536,768,1,1}; do
    for a in 0 1 2 4 6 7 8 16; do echo "-- $shape ablate $a: $(ONLY=$shape Y7T_CONV_ABLATE=$a timeout 100 python scripts/bench_conv.py 32 100 2>&1 | grep -v amdgpu | grep ' 1/1 ' | cut -c1-70)"; done
  done | tee -a $O/summary.txt
  ;;

exp_r4misc)
  say "exp_r4misc a: parity of the experiments -- stem with full-line stores (Y7T_STEM_LINES=1), LDS max-pool (default), 64-channel tiles instead of split-K (Y7T_CONV_NARROW=1)"
  Y7T_STEM_LINES=1 timeout 300 python -m pytest tests/test_detector_gpu.py -q -m gpu -k "fused_stem or whole_network" > $O/t_stemlines.log 2>&1; echo "rc=$?" >> $O/t_stemlines.log; tailsum $O/t_stemlines.log 2
  Y7T_STEM_LINES=1 Y7T_CONV_NARROW=1 timeout 500 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -k "every_op or launch_list" > $O/t_misc_pinned.log 2>&1; echo "rc=$?" >> $O/t_misc_pinned.log; tailsum $O/t_misc_pinned.log 2
  say "exp_r4misc b: per-op tables of the launch list, one variant per run (same session)"
  for v in default:X=1 stemlines:Y7T_STEM_LINES=1 narrow:Y7T_CONV_NARROW=1 poolgeneric:Y7T_POOL_LDS=0 default2:X=1; do
    n=${v%%:*}; e=${v#*:}; env $e NAME=$n OUT=$O bash scripts/per_layer_table.sh | tee -a $O/summary.txt
  done
  python3 - $O <<'PY' | tee -a $O/summary.txt
import sys, re
o = sys.argv[1]
def rows(n):
    r = {}
    try:
        for l in open("%s/per_layer_%s.txt" % (o, n)):
            m = re.match(r"\s*(\d+) (.+?)\s+(\d+x\d+)\s+(\d+)->(\d+)\s+(\d)/(\d)\s+([\d.]+) us", l)
            if m: r[int(m.group(1))] = (m.group(2).strip(), m.group(3), m.group(4), m.group(5), m.group(6) + "/" + m.group(7), float(m.group(8)))
    except Exception as e:
        print("no table", n, e)
    return r
base = rows("default")
for n in ("stemlines", "narrow", "poolgeneric", "default2"):
    v = rows(n)
    if not v or not base: continue
    diff = [(i, base[i], v[i]) for i in base if i in v and (abs(v[i][5] - base[i][5]) > 0.04 * base[i][5] or base[i][0] != v[i][0])]
    print("-- %s vs default: list %.1f -> %.1f us" % (n, sum(x[5] for x in base.values()), sum(x[5] for x in v.values())))
    for i, b, w in diff[:14]:
        print("   op %3d %-38s %9s %4s->%-4s %s  %8.1f -> %8.1f us  (%s)" % (i, b[0], b[1], b[2], b[3], b[4], b[5], w[5], w[0]))
PY
  ;;

exp_ws_s2)
  say "exp_ws_s2 a: the stride-2 weights-stationary kernel for the 640^2 64 -> 128 layer (csrc/y7t_conv_ws_s2.hip): layer parity vs torch fp32, then inside the benchmarked list"
  timeout 400 python -m pytest tests/test_detector_gpu.py -q -m gpu -k "stride2_weights_stationary" > $O/t_ws_s2.log 2>&1; echo "rc=$?" >> $O/t_ws_s2.log; tailsum $O/t_ws_s2.log 3
  timeout 500 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -k "every_op or launch_list" > $O/t_ws_s2_pinned.log 2>&1; echo "rc=$?" >> $O/t_ws_s2_pinned.log; tailsum $O/t_ws_s2_pinned.log 3
  say "exp_ws_s2 b: the layer alone at 32 frames: generic kernel (Y7T_CONV_WS_S2=0) vs ws_s2, with 256 / 248 / 512 workgroups"
  for v in "Y7T_CONV_WS_S2=0" "X=1" "Y7T_CONV_WS_WGS=248" "Y7T_CONV_WS_WGS=512"; do echo "-- $v: $(env $v ONLY=640,64,128,3,2 timeout 120 python scripts/bench_conv.py 32 50 2>&1 | grep ' 3/2 ' | cut -c1-70)"; done | tee -a $O/summary.txt
  say "exp_ws_s2 c: bench lines: with, without (Y7T_CONV_WS_S2=0), with"
  X="--steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode"
  timeout 300 python bench.py $X > $O/bench_ws2.json 2> $O/bench_ws2.err
  Y7T_CONV_WS_S2=0 timeout 300 python bench.py $X > $O/bench_nows2.json 2> $O/bench_nows2.err
  timeout 300 python bench.py $X > $O/bench_ws2b.json 2> $O/bench_ws2b.err
  benchsum ws2 nows2 ws2b
  ;;

exp_lowering)
  say "exp_lowering: round 4's lowering rules for the small maps (3x3 on the strip kernel from 200 workgroups, 64-row panels below 256 workgroups / 500 1x1 tiles) against round 3's thresholds: per-op tables + bench lines"
  R3="Y7T_CONV_PATCH_MIN_PIX=65536 Y7T_CONV_PATCH_PANEL64_BELOW=0 Y7T_CONV_1X1_PANEL64_BELOW=0"
  NAME=low_default OUT=$O bash scripts/per_layer_table.sh | tee -a $O/summary.txt
  env $R3 NAME=low_r3 OUT=$O bash scripts/per_layer_table.sh | tee -a $O/summary.txt
  X="--steps 20 --warmup 5 --no_cpu_baseline"
  timeout 400 python bench.py $X > $O/bench_low.json 2> $O/bench_low.err
  env $R3 timeout 400 python bench.py $X > $O/bench_lowr3.json 2> $O/bench_lowr3.err
  timeout 400 python bench.py $X > $O/bench_low2.json 2> $O/bench_low2.err
  benchsum low lowr3 low2
  ;;

exp_spp3)
  say "exp_spp3: the three SPPCSPC pools as one launch (k_spp3_lds): whole-network + teacher-forced parity, per-op table"
  timeout 400 python -m pytest tests/test_detector_gpu.py -q -m gpu -k "whole_network or upsample_on_read" > $O/t_spp3.log 2>&1; echo "rc=$?" >> $O/t_spp3.log; tailsum $O/t_spp3.log 2
  timeout 500 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -k "every_op or launch_list" > $O/t_spp3_pinned.log 2>&1; echo "rc=$?" >> $O/t_spp3_pinned.log; tailsum $O/t_spp3_pinned.log 2
  NAME=spp3 OUT=$O bash scripts/per_layer_table.sh | tee -a $O/summary.txt
  Y7T_SPP3=0 NAME=nospp3 OUT=$O bash scripts/per_layer_table.sh | tee -a $O/summary.txt
  grep -h "maxpool\|spp3\|ws_s2\| 0 stem\|TOTAL" $O/per_layer_spp3.txt $O/per_layer_nospp3.txt | cut -c1-120 | tee -a $O/summary.txt
  ;;

r5_chained)
  say "r5_chained: the chained detect -> NMS -> ByteTrack parity run (tests/test_chained_gpu.py, VERDICT r4 next 1)"
  timeout 900 python -m pytest -x -q -m gpu -s tests/test_chained_gpu.py > $O/t_chained.log 2>&1; echo "rc=$?" >> $O/t_chained.log
  grep -h "hand-over of\|graded against" $O/t_chained.log | cut -c1-700 | tee -a $O/summary.txt; tailsum $O/t_chained.log 6
  ;;

r5_wgs)
  say "r5_wgs: persistent kernels over-decomposed (Y7T_CONV_WS_WGS = 256 default / 512 / 1024: the hardware dispatcher as the dynamic scheduler), bench lines in one session"
  X="--steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode"
  for w in 256 512 1024 256; do
    Y7T_CONV_WS_WGS=$w timeout 300 python bench.py $X > $O/bench_wgs$w.json 2> $O/bench_wgs$w.err; benchsum wgs$w
  done
  ;;

r5_dyn)
  say "r5_dyn a: parity -- ws64 / ws128 layer cases (static partition through the single-layer entry point), the tile-counter forms inside a plan, then the benchmarked list op by op (ws64 dyn; and with Y7T_CONV_WS128=1)"
  timeout 600 python -m pytest tests/test_detector_gpu.py -q -m gpu -k "weights_stationary" > $O/t_dyn_layers.log 2>&1; echo "rc=$?" >> $O/t_dyn_layers.log; tailsum $O/t_dyn_layers.log 3
  timeout 600 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -k "every_op or launch_list" > $O/t_dyn_pinned.log 2>&1; echo "rc=$?" >> $O/t_dyn_pinned.log; tailsum $O/t_dyn_pinned.log 3
  Y7T_CONV_WS128=1 timeout 600 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -k "every_op" > $O/t_dyn_pinned128.log 2>&1; echo "rc=$?" >> $O/t_dyn_pinned128.log; tailsum $O/t_dyn_pinned128.log 3
  say "r5_dyn b: bench lines in one session: ws64 on the tile counter (default) | static (Y7T_CONV_WS_DYN=0) | + ws128 on the counter | + ws128 static | default again"
  X="--steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode"
  for v in dyn:X=1 static:Y7T_CONV_WS_DYN=0 dyn128:Y7T_CONV_WS128=1 static128:Y7T_CONV_WS128=1,Y7T_CONV_WS_DYN=0 dyn2:X=1 dyn128b:Y7T_CONV_WS128=1; do
    n=${v%%:*}; e=${v#*:}; env ${e//,/ } timeout 300 python bench.py $X > $O/bench_$n.json 2> $O/bench_$n.err; benchsum $n
  done
  say "r5_dyn c: per-op tables of the launch list alone on the chip (through the plan: tile counters live): default, static, + ws128"
  for v in dyn:X=1 static:Y7T_CONV_WS_DYN=0 dyn128:Y7T_CONV_WS128=1; do
    n=${v%%:*}; e=${v#*:}; env ${e//,/ } NAME=$n OUT=$O bash scripts/per_layer_table.sh | tee -a $O/summary.txt
  done
  grep -h "ws64\|ws128\|patch<16,16,128>\|TOTAL\|total" $O/per_layer_dyn.txt $O/per_layer_static.txt $O/per_layer_dyn128.txt 2>/dev/null | cut -c1-200 | tee -a $O/summary.txt
  ;;

r5_batch)
  say "r5_batch: frames per step (bench.py --batch): tile quantisation of the 4-per-CU kernels against the batch (25 pixel tiles per 80 x 80 frame: 1600 workgroups on 512 slots at 32 frames)"
  X="--steps 12 --warmup 4 --no_cpu_baseline --no_latency_mode --no_other_workloads"
  for b in ${BATCHES:-32 24 40 48 64 32}; do
    timeout 400 python bench.py $X --batch $b > $O/bench_b$b.json 2> $O/bench_b$b.err; benchsum b$b
  done
  ;;

r5_latency)
  say "r5_latency: batch-1 latency mode (reference Timer semantics), the frame's H2D copy as 1 / 2 / 4 concurrent pieces, one session"
  for n in 1 2 4 1 2; do
    timeout 400 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_other_workloads --upload_pieces $n > $O/bench_up$n.json 2> $O/bench_up$n.err
    python3 - $O/bench_up$n.json $n <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    l = json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
    print("upload pieces %s: latency mode %s" % (sys.argv[2], {k: v for k, v in l["latency_mode"].items() if k != "note"}))
except Exception as e:
    print("upload pieces %s: no line (%r)" % (sys.argv[2], e))
PY
  done
  ;;

r5_b40)
  say "r5_b40: the benchmarked configuration at 40 frames per forward: the pinned parity tests (launch list, every op teacher-forced, heads, candidates, boxes, all levels undamped), the tile-counter plan test"
  timeout 1500 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -s -x > $O/t_b40_pinned.log 2>&1; echo "rc=$?" >> $O/t_b40_pinned.log
  grep -h "launch list @\|candidates:\|oracle keeps" $O/t_b40_pinned.log | cut -c1-500 | tee -a $O/summary.txt; tailsum $O/t_b40_pinned.log 3
  timeout 600 python -m pytest tests/test_detector_gpu.py -q -m gpu -k "tile_counter or weights_stationary" > $O/t_b40_ws.log 2>&1; echo "rc=$?" >> $O/t_b40_ws.log; tailsum $O/t_b40_ws.log 2
  ;;

r5_branches)
  say "r5_branches a: parity with the Detect branches side by side (product library, default): the benchmarked list op by op, candidates, boxes"
  timeout 1200 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -x -k "launch_list or every_op or candidates_before or boxes_end_to_end" > $O/t_branches.log 2>&1; echo "rc=$?" >> $O/t_branches.log; tailsum $O/t_branches.log 3
  say "r5_branches b: bench lines, measuring build (same kernels; it reads the experiment switches): Detect branches side by side (1) / in list order (0), alternating; then the small-map panel rules at 40 frames"
  X="--steps 16 --warmup 4 --no_cpu_baseline --no_latency_mode --no_other_workloads"
  export Y7T_LIB=$LIBD/liby7t_ablate.so
  for v in br1:Y7T_DETECT_BRANCHES=1 br0:Y7T_DETECT_BRANCHES=0 br1b:Y7T_DETECT_BRANCHES=1 br0b:Y7T_DETECT_BRANCHES=0 p128:Y7T_CONV_PATCH_PANEL64_BELOW=0 p64:Y7T_CONV_PATCH_PANEL64_BELOW=400 x64:Y7T_CONV_1X1_PANEL64_BELOW=600 x128:Y7T_CONV_1X1_PANEL64_BELOW=0 br1c:Y7T_DETECT_BRANCHES=1; do
    n=${v%%:*}; e=${v#*:}; env ${e//,/ } timeout 300 python bench.py $X > $O/bench_$n.json 2> $O/bench_$n.err; benchsum $n
  done
  unset Y7T_LIB
  ;;

r5_panel240)
  say "r5_panel240: 128-row panels for the six 20 x 20 512 -> 512 layers only (Y7T_CONV_PATCH_PANEL64_BELOW=240; measuring build), bench lines alternating + per-op tables"
  X="--steps 16 --warmup 4 --no_cpu_baseline --no_latency_mode --no_other_workloads"
  export Y7T_LIB=$LIBD/liby7t_ablate.so
  for v in base:X=1 p240:Y7T_CONV_PATCH_PANEL64_BELOW=240 base2:X=1 p240b:Y7T_CONV_PATCH_PANEL64_BELOW=240; do
    n=${v%%:*}; e=${v#*:}; env ${e//,/ } timeout 300 python bench.py $X > $O/bench_$n.json 2> $O/bench_$n.err; benchsum $n
  done
  for v in base:X=1 p240:Y7T_CONV_PATCH_PANEL64_BELOW=240; do
    n=${v%%:*}; e=${v#*:}; env ${e//,/ } NAME=$n OUT=$O bash scripts/per_layer_table.sh | tee -a $O/summary.txt
  done
  grep -h " 20x20 .*512->512\|TOTAL" $O/per_layer_base.txt $O/per_layer_p240.txt 2>/dev/null | cut -c1-160 | tee -a $O/summary.txt
  unset Y7T_LIB
  ;;

suite_all)
  say "suite_all: python -m pytest tests/ -q -m gpu (no -x)"
  timeout 1500 python -m pytest tests/ -q -m gpu > $O/t_suite.log 2>&1; echo "rc=$?" >> $O/t_suite.log; tailsum $O/t_suite.log 6
  ;;

r5_ws128s2)
  say "r5_ws128s2 a: the stride-2 form of ws128 (csrc/y7t_conv_ws128.hip, S2): layer parity vs torch fp32 (static partition), then inside the benchmarked list op by op (tile counter; measuring build so that the lowering honours Y7T_CONV_WS128_S2)"
  timeout 600 python -m pytest tests/test_detector_gpu.py -q -m gpu -k "weights_stationary_128" > $O/t_s2_layers.log 2>&1; echo "rc=$?" >> $O/t_s2_layers.log; tailsum $O/t_s2_layers.log 3
  export Y7T_LIB=$LIBD/liby7t_ablate.so
  Y7T_CONV_WS128_S2=1 timeout 900 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -s -k "every_op or launch_list" > $O/t_s2_pinned.log 2>&1; echo "rc=$?" >> $O/t_s2_pinned.log
  grep -h "launch list @" $O/t_s2_pinned.log | cut -c1-400 | tee -a $O/summary.txt; tailsum $O/t_s2_pinned.log 3
  say "r5_ws128s2 b: bench lines, alternating (0 = patch_s2 for those two layers, 1 = ws128_s2)"
  X="--steps 16 --warmup 4 --no_cpu_baseline --no_latency_mode --no_other_workloads"
  for v in s2off:Y7T_CONV_WS128_S2=0 s2on:Y7T_CONV_WS128_S2=1 s2offb:Y7T_CONV_WS128_S2=0 s2onb:Y7T_CONV_WS128_S2=1; do
    n=${v%%:*}; e=${v#*:}; env ${e//,/ } timeout 300 python bench.py $X > $O/bench_$n.json 2> $O/bench_$n.err; benchsum $n
  done
  say "r5_ws128s2 c: per-op tables"
  for v in s2off:Y7T_CONV_WS128_S2=0 s2on:Y7T_CONV_WS128_S2=1; do
    n=${v%%:*}; e=${v#*:}; env ${e//,/ } NAME=$n OUT=$O bash scripts/per_layer_table.sh | tee -a $O/summary.txt
  done
  grep -h " 128->256  3/2\|TOTAL" $O/per_layer_s2off.txt $O/per_layer_s2on.txt 2>/dev/null | cut -c1-160 | tee -a $O/summary.txt
  unset Y7T_LIB
  ;;

r5_coop)
  say "r5_coop a: sparse association -- row contexts of the cost pass handed out through the scalar registers, large components on a wave with their state in registers: device parity (goldens, random scenes, cfg3 at full size vs the oracle, DeepSORT)"
  timeout 900 python -m pytest tests/test_tracker_gpu.py tests/test_fullsize_gpu.py tests/test_reid_gpu.py -q -m gpu -k "not w6_1280 and not conv_layer and not nms_output" > $O/t_coop.log 2>&1; echo "rc=$?" >> $O/t_coop.log; tailsum $O/t_coop.log 4
  say "r5_coop b: the frame step alone (scripts/time_tracker.py): previous library, this one"
  for v in prev:$LIBD/liby7t_prev.so new:$LIBD/liby7t.so; do
    n=${v%%:*}; l=${v#*:}; echo "-- $n" | tee -a $O/summary.txt
    Y7T_LIB=$l timeout 300 python scripts/time_tracker.py > $O/time_tracker_$n.txt 2>&1; grep -h "n_obj=500\|sparse association" $O/time_tracker_$n.txt | grep -v "threads=64 \|threads=256 " | cut -c1-300 | tee -a $O/summary.txt
  done
  say "r5_coop c: cfg3 (BoT-SORT, 500 objects) and the headline, previous / this library alternating"
  X="--steps 10 --warmup 3 --no_cpu_baseline --no_latency_mode --no_other_workloads"
  for v in prev:$LIBD/liby7t_prev.so new:$LIBD/liby7t.so prevb:$LIBD/liby7t_prev.so newb:$LIBD/liby7t.so; do
    n=${v%%:*}; l=${v#*:}; Y7T_LIB=$l timeout 300 python bench.py $X --workload cfg3 > $O/bench_cfg3_$n.json 2> $O/bench_cfg3_$n.err; benchsum cfg3_$n
  done
  for v in prev:$LIBD/liby7t_prev.so new:$LIBD/liby7t.so; do
    n=${v%%:*}; l=${v#*:}; Y7T_LIB=$l timeout 300 python bench.py --steps 16 --warmup 4 --no_cpu_baseline --no_latency_mode --no_other_workloads > $O/bench_cfg2_$n.json 2> $O/bench_cfg2_$n.err; benchsum cfg2_$n
  done
  timeout 300 python bench.py $X --workload cfg4 > $O/bench_cfg4_new.json 2> $O/bench_cfg4_new.err; benchsum cfg4_new
  ;;

r5_coop2)
  say "r5_coop2: the wave solve's threshold (components of more than N rows go to a wave: 8 = the product library, 4, 2 = variant builds of y7t_tracker.hip with -DY7T_COOP_MIN=N) + the overlap pre-test of the IoU cost pass; parity of the product library first"
  timeout 900 python -m pytest tests/test_tracker_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "not w6_1280 and not conv_layer and not nms_output" > $O/t_coop2.log 2>&1; echo "rc=$?" >> $O/t_coop2.log; tailsum $O/t_coop2.log 4
  for v in ${LIBS:-prev:liby7t_prev.so min8:liby7t.so min4:liby7t_coop4.so min2:liby7t_coop2.so}; do
    n=${v%%:*}; l=$LIBD/${v#*:}; [ -f $l ] || continue; echo "-- $n" | tee -a $O/summary.txt
    Y7T_LIB=$l timeout 300 python scripts/time_tracker.py > $O/time_tracker_$n.txt 2>&1; grep -h "n_obj=500\|sparse association\|large components" $O/time_tracker_$n.txt | grep -v "threads=64 \|threads=256 " | cut -c1-300 | tee -a $O/summary.txt
  done
  X="--steps 10 --warmup 3 --no_cpu_baseline --no_latency_mode --no_other_workloads"
  for v in ${LIBS:-prev:liby7t_prev.so min8:liby7t.so min4:liby7t_coop4.so min2:liby7t_coop2.so}; do
    n=${v%%:*}; l=$LIBD/${v#*:}; [ -f $l ] || continue
    Y7T_LIB=$l timeout 300 python bench.py $X --workload cfg3 > $O/bench_cfg3_$n.json 2> $O/bench_cfg3_$n.err; benchsum cfg3_$n
  done
  ;;

r5_tracker_ab)
  say "r5_tracker_ab: the tracker of this commit against a previous build of the library (lib/liby7t_prev.so: copy one there first): device parity, the frame step alone (ByteTrack 80 / 500 objects, DeepSORT), cfg3 / cfg2 / cfg4 alternating"
  timeout 900 python -m pytest tests/test_tracker_gpu.py tests/test_fullsize_gpu.py tests/test_reid_gpu.py tests/test_cli_gpu.py tests/test_multirank_gpu.py -q -m gpu -k "not w6_1280 and not conv_layer and not nms_output" > $O/t_tracker.log 2>&1; echo "rc=$?" >> $O/t_tracker.log; tailsum $O/t_tracker.log 4
  for l in liby7t_prev.so liby7t.so; do [ -f $LIBD/$l ] || continue; echo "-- $l" | tee -a $O/summary.txt
    Y7T_LIB=$LIBD/$l timeout 200 python scripts/time_tracker.py 2>&1 | grep "threads=1024\|total\|sparse assoc\|large comp" | cut -c1-300 | tee -a $O/summary.txt
    Y7T_LIB=$LIBD/$l timeout 200 python scripts/time_deepsort.py 2>&1 | grep -A1 dim512 | cut -c1-200 | tee -a $O/summary.txt
  done
  X="--steps 10 --warmup 3 --no_cpu_baseline --no_latency_mode --no_other_workloads"
  for w in cfg3 cfg2 cfg4; do for v in prev:liby7t_prev.so new:liby7t.so prevb:liby7t_prev.so newb:liby7t.so; do
    n=${w}_${v%%:*}; l=$LIBD/${v#*:}; [ -f $l ] || continue
    Y7T_LIB=$l timeout 300 python bench.py $X --workload $w > $O/bench_$n.json 2> $O/bench_$n.err; benchsum $n
  done; done
  ;;

suite)
  say "suite: python -m pytest tests/ -x -q -m gpu"
  timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/t_suite.log 2>&1; echo "rc=$?" >> $O/t_suite.log; tailsum $O/t_suite.log 3
  ;;

bench)
  say "bench: the driver's line"
  timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "rc=$?" | tee -a $O/summary.txt
  cp $O/bench_line.json $O/bench_default.json; benchsum default
  ;;

bench_variants)
  say "bench_variants (one session): default, chaotic weights, tracker chain on reserved CUs"
  X="--steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode"
  timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  timeout 300 python bench.py $X --weights chaotic > $O/bench_chaotic.json 2> $O/bench_chaotic.err
  timeout 300 python bench.py $X --cu_reserve 8 > $O/bench_cu8.json 2> $O/bench_cu8.err
  timeout 300 python bench.py $X --cu_reserve 16 > $O/bench_cu16.json 2> $O/bench_cu16.err
  timeout 300 python bench.py $X --cu_reserve 8 --cu_reserve_nms 1 > $O/bench_cu8nms.json 2> $O/bench_cu8nms.err
  timeout 300 python bench.py $X > $O/bench_default2.json 2> $O/bench_default2.err
  benchsum default chaotic cu8 cu16 cu8nms default2
  for n in default chaotic cu8 cu16 cu8nms default2; do [ -s $O/bench_$n.err ] && { echo "-- stderr of $n"; grep -v amdgpu.ids $O/bench_$n.err | tail -4; }; done | tee -a $O/summary.txt
  ;;

stats)
  say "stats: rocprofv3 --kernel-trace --stats of the bench command (kernel stats only; the per-op table and the PMC passes are the profile step)"
  P=$O/prof; mkdir -p $P
  ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks
    timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $ROOT/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode --no_other_workloads > $P/bench_under_rocprof.log 2>&1
    f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/bench_kernel_stats.csv )
  head -14 $P/bench_kernel_stats.csv 2>/dev/null | cut -c1-160 | tee -a $O/summary.txt
  grep -o '"value": [0-9.]*, "unit": "frames/s"\|"launch_list_ms": [0-9.]*' $P/bench_under_rocprof.log | head -2 | tee -a $O/summary.txt
  ;;

closing_small)
  say "closing_small: the chained parity tests, smoke(), the driver's bench line"
  timeout 600 python -m pytest tests/test_chained_gpu.py -q -m gpu > $O/t_chained.log 2>&1; echo "rc=$?" >> $O/t_chained.log; tailsum $O/t_chained.log 3
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tailsum $O/smoke.log 2
  timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "rc=$?" | tee -a $O/summary.txt
  cp $O/bench_line.json $O/bench_default.json; benchsum default
  ;;

profile)
  say "profile: rocprofv3 of the bench command + the launch list alone (kernel stats, per-op table, HBM traffic, MFMA busy)"
  P=$O/prof; mkdir -p $P
  COMMIT=$(cat $ROOT/.commit_stamp 2>/dev/null || echo unknown)
  ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks /tmp/kt /tmp/pf /tmp/pw /tmp/mb
    timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $ROOT/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode --no_other_workloads > $P/bench_under_rocprof.log 2>&1
    f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/bench_kernel_stats.csv
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $ROOT/scripts/forward_only.py 4 > $P/forward_only.log 2>&1
    f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $P/forward_kernel_trace.csv
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python $ROOT/scripts/forward_only.py 3 > /tmp/pf.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python $ROOT/scripts/forward_only.py 3 > /tmp/pw.log 2>&1
    CTRS="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
    timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/mb -- python $ROOT/scripts/forward_only.py 3 > /tmp/mb.log 2>&1
    python3 $ROOT/scripts/profile_reduce.py $P "$CTRS" "$COMMIT" 2>&1 | tee -a $O/summary.txt )
  head -14 $P/bench_kernel_stats.csv 2>/dev/null | cut -c1-160 | tee -a $O/summary.txt
  grep -o '"value": [0-9.]*, "unit": "frames/s"\|"launch_list_ms": [0-9.]*' $P/bench_under_rocprof.log | head -2 | tee -a $O/summary.txt
  ;;

exp_ws)
  say "exp_ws a: weights-stationary 64 -> 64 kernel (csrc/y7t_conv_ws.hip): layer parity vs torch fp32, then inside the benchmarked list (teacher-forced)"
  timeout 300 python -m pytest tests/test_detector_gpu.py -q -m gpu -k weights_stationary > $O/t_ws.log 2>&1; echo "rc=$?" >> $O/t_ws.log; tailsum $O/t_ws.log
  timeout 400 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -k "every_op or launch_list" > $O/t_ws_pinned.log 2>&1; echo "rc=$?" >> $O/t_ws_pinned.log; tailsum $O/t_ws_pinned.log
  say "exp_ws b: per-layer timing at 32 frames (64->64 3/1 rows): default (patch_mt / patch) vs ws64; ws64 with fewer / more workgroups than compute units"
  Y7T_CONV_WS=0 timeout 200 python scripts/bench_conv.py 32 > $O/b_default.txt 2>&1
  timeout 200 python scripts/bench_conv.py 32 > $O/b_ws.txt 2>&1
  Y7T_CONV_WS_WGS=512 timeout 200 python scripts/bench_conv.py 32 > $O/b_ws_512.txt 2>&1
  for f in default ws ws_512; do echo "-- $f"; grep "  64->64 \|TOTAL" $O/b_$f.txt; done | tee -a $O/summary.txt
  say "exp_ws c: bench line with it on"
  timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/bench_ws.json 2> $O/bench_ws.err
  Y7T_CONV_WS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/bench_nows.json 2> $O/bench_nows.err
  benchsum ws nows
  ;;

exp_ws4)
  say "exp_ws4 a: hand-scheduled weights-stationary kernel: layer parity vs torch fp32, then inside the benchmarked list (teacher-forced)"
  timeout 300 python -m pytest tests/test_detector_gpu.py -q -m gpu -k weights_stationary > $O/t_ws.log 2>&1; echo "rc=$?" >> $O/t_ws.log; tailsum $O/t_ws.log
  timeout 400 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -k "every_op or launch_list" > $O/t_ws_pinned.log 2>&1; echo "rc=$?" >> $O/t_ws_pinned.log; tailsum $O/t_ws_pinned.log
  say "exp_ws4 b: per-layer timing at 32 frames"
  timeout 200 python scripts/bench_conv.py 32 > $O/b_ws.txt 2>&1
  grep "  64->64 \|TOTAL" $O/b_ws.txt | tee -a $O/summary.txt
  say "exp_ws4 c: issue counters of the kernel"
  OUT=$O/pmc_ws bash scripts/pmc_kernel.sh c64_ws python scripts/forward_only.py 2 > $O/pmc_ws.txt 2>&1; grep -v "^pass" $O/pmc_ws/summary.txt | awk '{print $1, $2, $4}' | sort -u | tee -a $O/summary.txt
  say "exp_ws4 d: bench line"
  timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/bench_ws.json 2> $O/bench_ws.err
  benchsum ws
  ;;

exp_issue)
  say "exp_issue a: what one wave issues in the shadow of its MFMAs (scripts/ubench/issue_model.hip)"
  timeout 120 scripts/ubench/issue_model > $O/issue_model.txt 2>&1; cat $O/issue_model.txt | tee -a $O/summary.txt
  say "exp_issue b: ws64 with the accumulators in arch VGPRs (default) / in ACC registers (Y7T_WS_ACC=a): parity of the latter, per-layer time of both"
  Y7T_WS_ACC=a timeout 300 python -m pytest tests/test_detector_gpu.py -q -m gpu -k weights_stationary > $O/t_ws_acca.log 2>&1; echo "rc=$?" >> $O/t_ws_acca.log; tailsum $O/t_ws_acca.log
  for v in v a; do echo "-- acc=$v"; ONLY=320,64,64,3,1 Y7T_WS_ACC=$v timeout 100 python scripts/bench_conv.py 32 200 2>&1 | grep "64->64"; done | tee -a $O/summary.txt
  ;;

exp_noslp)
  say "exp_noslp a: conv translation units without SLP vectorisation (no packed-fp32 epilogue instructions) + ws64 with a plain-fp32 micro-program: parity"
  timeout 600 python -m pytest tests/test_detector_gpu.py -q -m gpu -x > $O/t_det.log 2>&1; echo "rc=$?" >> $O/t_det.log; tailsum $O/t_det.log
  timeout 600 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu > $O/t_pinned.log 2>&1; echo "rc=$?" >> $O/t_pinned.log; tailsum $O/t_pinned.log
  say "exp_noslp b: per-layer timing at 32 frames: previous library (lib/liby7t_prev.so) vs this one, same session"
  Y7T_LIB=$LIBD/liby7t_prev.so timeout 200 python scripts/bench_conv.py 32 > $O/b_prev.txt 2>&1
  timeout 200 python scripts/bench_conv.py 32 > $O/b_new.txt 2>&1
  paste <(cut -c1-52 $O/b_prev.txt) <(cut -c36-52 $O/b_new.txt) | tee -a $O/summary.txt
  say "exp_noslp c: bench lines, previous library then this one"
  Y7T_LIB=$LIBD/liby7t_prev.so timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/bench_prev.json 2> $O/bench_prev.err
  timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/bench_new.json 2> $O/bench_new.err
  benchsum prev new
  ;;

exp_clock)
  say "exp_clock a: shader clock per kernel of the per-layer benchmark at 32 frames (GRBM_GUI_ACTIVE / 8 / duration)"
  OUT=$O bash scripts/clock_probe.sh python scripts/bench_conv.py 32 10 > $O/clock_probe.log 2>&1; tail -40 $O/clock_probe.log | cut -c1-110 | tee -a $O/summary.txt
  say "exp_clock b: issue counters of the ws64 kernel on the 320 x 320 layer at 32 frames"
  ONLY=320,64,64,3,1 OUT=$O/pmc_ws32 bash scripts/pmc_kernel.sh c64_ws python scripts/bench_conv.py 32 2 > $O/pmc_ws32.txt 2>&1; awk '{print $(NF-2), $NF}' $O/pmc_ws32/summary.txt | tee -a $O/summary.txt
  say "exp_clock c: what the matrix pipe sustains on random operands by operand ORDER (scripts/ubench/mfma_patterns.hip)"
  timeout 100 scripts/ubench/mfma_patterns 2>&1 | tee -a $O/summary.txt
  ;;

exp_wsprobe)
  say "exp_wsprobe: the ws64 layer by data (operand power), layout (dense / slices of wider tensors) and working set"
  timeout 300 python scripts/ws_probe.py > $O/ws_probe.txt 2>&1; grep -v amdgpu.ids $O/ws_probe.txt | tee -a $O/summary.txt
  ;;

exp_wsabl)
  say "exp_wsabl: timing ablations of the ws64 kernel (wrong results): 1 no epilogue, 2 no pieces, 4 no fragment reads, 8 no wait / barrier, 16 no stores; 320 x 320, 32 frames"
  for a in ${ABLS:-0 1 2 3 4 8 16 18}; do echo "-- ablate $a"; Y7T_WS_ABLATE=$a QUICK=1 timeout 100 python scripts/ws_probe.py 2>&1 | grep -v amdgpu.ids; done | tee -a $O/summary.txt
  ;;

exp_latency)
  say "exp_latency a: per-layer timing of the launch list at ONE frame"
  timeout 200 python scripts/bench_conv.py 1 50 > $O/b_one_frame.txt 2>&1; grep -v amdgpu.ids $O/b_one_frame.txt | tee -a $O/summary.txt
  say "exp_latency b: kernel trace of the batch-1 latency mode (uint8 host frames, hipGraph replay)"
  OUT=$O bash scripts/latency_trace.sh > $O/latency_trace.log 2>&1; cat $O/latency_trace.txt | cut -c1-120 | tee -a $O/summary.txt; tail -3 $O/latency_mode_under_rocprof.log | tee -a $O/summary.txt
  ;;

exp_arena)
  say "exp_arena a: the tracker's index lists in LDS for the length of a frames launch: tracker tests on the device"
  timeout 600 python -m pytest tests/test_tracker_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x > $O/t_trk.log 2>&1; echo "rc=$?" >> $O/t_trk.log; tailsum $O/t_trk.log
  say "exp_arena b: bench lines (cfg2, cfg3) with the arena (default) and without (Y7T_TRACKER_ARENA=0), same session"
  for w in cfg2 cfg3; do for a in 1 0; do
    Y7T_TRACKER_ARENA=$a timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode --workload $w > $O/bench_${w}_arena$a.json 2> $O/bench_${w}_arena$a.err
  done; done
  benchsum cfg2_arena1 cfg2_arena0 cfg3_arena1 cfg3_arena0
  ;;

pmc_queues)
  say "pmc_queues: TA / TCP / TCC / SQ counters of the 1x1 (80x80 1024->512), stride-2 (160x160 256->512) generic layers and the patch kernel (80x80 256->256)"
  OUT=$O/pmc_raw SHAPES="${SHAPES:-1x1 s2 patch}" bash scripts/pmc_queues.sh > $O/pmc_queues.txt 2>&1
  grep -v "^  TCP_\|^  TA_\|^  TCC_\|^  SQ_\|^  GRBM" $O/pmc_queues.txt | tail -60 | tee -a $O/summary.txt
  ;;

*) say "unknown step $step";;
esac; done
say "done: $*"
