#!/bin/bash
# Round 3, FIRST GPU call: the experiments that were written and host-validated at the end of round 2 without GPU minutes (stride-2 patch kernel, 8-wave
# instances of the generic kernel, split-K fix-up, tracker candidate lists in LDS, DMA-late order of the stride-1 patch kernel).
#
#   BEFORE calling gpurun (here, on the CPU):   python -m yolov7_tracker_amd.build && python scripts/ablate/build_experiments.py fixup next
#   (the experimental libraries lib/exp_{fixup,next}.so travel with the snapshot; the stride-2 patch kernel and the 8-wave instances of the generic kernel
#   are in liby7t.so, opt-in by environment)
#   then:   gpurun --timeout 1500 -- 'bash scripts/gpu_r3a.sh'
#
# Every step is wrapped in its own timeout and writes to gpurun_out/r3a/; a failing experiment does not stop the others.
# What to do with the answers: DESIGN.md section 7 ("what the first GPU call of round 3 decides").
O=gpurun_out/r3a; mkdir -p $O
LIBD=$GRAFT_REPO_ROOT/yolov7-tracker_amd/lib
say() { echo "=== $*" | tee -a $O/summary.txt; }
# STEPS="0 1 2 3 4 5" (default: all, about 40 GPU-minutes); e.g. STEPS="0 1 2" bash scripts/gpu_r3a.sh for the convolution experiments only (about 25)
STEPS=${STEPS:-"0 1 2 3 4 5"}
want() { [[ " $STEPS " == *" $1 "* ]]; }

# ---- 0. the default build is what round 2 verified: layer tests + the pinned launch list (2 min) ----
if want 0; then
say "0. default suite (conv layers, pinned list)"
timeout 200 python -m pytest tests/test_detector_gpu.py tests/test_detector_pinned_gpu.py -x -q -m gpu > $O/t0_default.log 2>&1; echo "rc=$?" >> $O/t0_default.log
tail -2 $O/t0_default.log | tee -a $O/summary.txt
fi

# ---- 1. stride-2 LDS-patch kernel (csrc/y7t_conv_patch_s2.hip; korder 4) ----
if want 1; then
say "1a. stride-2 patch kernel: layer parity vs torch fp32"
Y7T_TEST_EXPERIMENTS=1 timeout 150 python -m pytest tests/test_detector_gpu.py -q -m gpu -k stride2 > $O/t1a_s2_layers.log 2>&1; echo "rc=$?" >> $O/t1a_s2_layers.log
tail -2 $O/t1a_s2_layers.log | tee -a $O/summary.txt
Y7T_TEST_EXPERIMENTS=1 Y7T_CONV_PATCH_S2_NW=8 timeout 150 python -m pytest tests/test_detector_gpu.py -q -m gpu -k stride2 > $O/t1a_s2_layers_nw8.log 2>&1; echo "rc=$?" >> $O/t1a_s2_layers_nw8.log
echo "512-thread form:" | tee -a $O/summary.txt; tail -2 $O/t1a_s2_layers_nw8.log | tee -a $O/summary.txt
say "1b. stride-2 patch kernel inside the benchmarked launch list, teacher-forced against the oracle"
Y7T_CONV_PATCH_S2=1 timeout 250 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu > $O/t1b_s2_pinned.log 2>&1; echo "rc=$?" >> $O/t1b_s2_pinned.log
tail -2 $O/t1b_s2_pinned.log | tee -a $O/summary.txt
say "1c. per-layer timing, 32 frames: generic vs patch_s2 (256-channel panels where Cout allows) vs 128-channel panels only vs the 512-thread form (16x16 pixels, one workgroup per CU)"
timeout 200 python scripts/bench_conv.py 32 > $O/b1c_default.txt 2>&1
Y7T_CONV_PATCH_S2=1 timeout 200 python scripts/bench_conv.py 32 > $O/b1c_s2.txt 2>&1
Y7T_CONV_PATCH_S2=1 Y7T_CONV_PATCH_S2_BN=128 timeout 200 python scripts/bench_conv.py 32 > $O/b1c_s2_bn128.txt 2>&1
Y7T_CONV_PATCH_S2=1 Y7T_CONV_PATCH_S2_NW=8 timeout 200 python scripts/bench_conv.py 32 > $O/b1c_s2_nw8.txt 2>&1
Y7T_CONV_PATCH_S2=1 Y7T_CONV_PATCH_S2_ORDER=1 timeout 200 python scripts/bench_conv.py 32 > $O/b1c_s2_late.txt 2>&1                            # DMAs behind the step's MFMAs
Y7T_CONV_PATCH_S2=1 Y7T_CONV_PATCH_S2_NW=8 Y7T_CONV_PATCH_S2_ORDER=1 timeout 200 python scripts/bench_conv.py 32 > $O/b1c_s2_nw8_late.txt 2>&1
for f in default s2 s2_bn128 s2_nw8 s2_late s2_nw8_late; do echo "-- $f"; grep " 3/2 \|TOTAL" $O/b1c_$f.txt; done | tee -a $O/summary.txt
Y7T_TEST_EXPERIMENTS=1 Y7T_CONV_PATCH_S2_ORDER=1 timeout 150 python -m pytest tests/test_detector_gpu.py -q -m gpu -k stride2 > $O/t1c_s2_layers_late.log 2>&1; echo "rc=$?" >> $O/t1c_s2_layers_late.log
echo "dma-late order, parity:" | tee -a $O/summary.txt; tail -2 $O/t1c_s2_layers_late.log | tee -a $O/summary.txt
say "1d. bench line with the stride-2 kernel on (all eight layers / only the layers with 256-channel panels)"
timeout 240 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
Y7T_CONV_PATCH_S2=1 timeout 240 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/bench_s2.json 2> $O/bench_s2.err
Y7T_CONV_PATCH_S2=1 Y7T_CONV_PATCH_S2_MIN_COUT=256 timeout 240 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/bench_s2_wide.json 2> $O/bench_s2_wide.err
Y7T_CONV_PATCH_S2=1 Y7T_CONV_PATCH_S2_NW=8 timeout 240 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/bench_s2_nw8.json 2> $O/bench_s2_nw8.err
python - <<'PY' | tee -a $O/summary.txt
import json
for n in ("default", "s2", "s2_wide", "s2_nw8"):
    try:
        l = json.loads(open("gpurun_out/r3a/bench_%s.json" % n).read().strip().splitlines()[-1])
        wc = l.get("parity", {}).get("well_conditioned", {})
        print("%-8s %.0f fps  list %.2f ms  frac %.4f  well-conditioned boxes matched %s / %s" % (n, l["value"], l["roofline"].get("launch_list_ms", float("nan")), l["roofline"]["frac"],
              wc.get("boxes_matched_same_class_1px_conf5e-3"), wc.get("boxes_oracle")))
    except Exception as e:
        print(n, "no bench line:", e)
PY
fi

# ---- 2. 8-wave instances of the generic kernel (512 threads, 256x256x64 / 256x128x64 tiles at two waves per SIMD), inside the default library ----
if want 2; then
# Y7T_CONV_NW8=<form> in the environment: the C dispatcher takes the eligible layers to the 8-wave tiles, and detector/graph.py::nw8_eligible lowers exactly
# those 1x1 layers with row-major weights (the weight panels are packed for the 128x32 tile); every other layer keeps its packing
say "2a. 8-wave instances in the benchmarked launch list, teacher-forced against the oracle"
Y7T_CONV_NW8=1 timeout 250 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu > $O/t2a_nw8_pinned.log 2>&1; echo "rc=$?" >> $O/t2a_nw8_pinned.log
tail -2 $O/t2a_nw8_pinned.log | tee -a $O/summary.txt
say "2b. per-layer timing (baseline: 1c's default table): 8 waves at 256x256x64 (1), 256x256x32 with a four-stage ring (2), 256x128 (6), 128x128 (7)"
[ -f $O/b1c_default.txt ] || timeout 200 python scripts/bench_conv.py 32 > $O/b1c_default.txt 2>&1
cp $O/b1c_default.txt $O/b2_default.txt
for v in 1 2 6 7; do Y7T_CONV_NW8=$v timeout 200 python scripts/bench_conv.py 32 > $O/b2_nw8_$v.txt 2>&1; done
for f in default nw8_1 nw8_2 nw8_6 nw8_7; do echo "-- $f"; grep "TOTAL\| 1/1 \| 3/2 \| 20x20 " $O/b2_$f.txt | head -60; done | tee -a $O/summary.txt
say "2c. bench line with the 8-wave instances on (alone, and together with the 512-thread stride-2 patch kernel)"
Y7T_CONV_NW8=1 timeout 240 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/bench_nw8.json 2> $O/bench_nw8.err
Y7T_CONV_NW8=1 Y7T_CONV_PATCH_S2=1 Y7T_CONV_PATCH_S2_NW=8 timeout 240 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_latency_mode > $O/bench_nw8_s2.json 2> $O/bench_nw8_s2.err
python - <<'PY' | tee -a $O/summary.txt
import json
for n in ("nw8", "nw8_s2"):
    try:
        l = json.loads(open("gpurun_out/r3a/bench_%s.json" % n).read().strip().splitlines()[-1])
        wc = l.get("parity", {}).get("well_conditioned", {})
        print("%-8s %.0f fps  list %.2f ms  frac %.4f  well-conditioned boxes matched %s / %s" % (n, l["value"], l["roofline"].get("launch_list_ms", float("nan")), l["roofline"]["frac"],
              wc.get("boxes_matched_same_class_1px_conf5e-3"), wc.get("boxes_oracle")))
    except Exception as e:
        print(n, "no bench line:", e)
PY
fi

# ---- 3. fixup: split-K reduced by the last arriving workgroup (batch-1 latency mode) ----
if want 3; then
say "3. fixup (Y7T_SPLITK_FIXUP, Y7T_CONV_SPLITK=2): layer parity incl. repeated launches, then latency mode against the default library"
if [ -f $LIBD/exp_fixup.so ]; then
  Y7T_LIB=$LIBD/exp_fixup.so Y7T_CONV_SPLITK=2 timeout 200 python -m pytest tests/test_detector_gpu.py -q -m gpu -k "conv_layer or whole_network or batch" > $O/t3_fixup.log 2>&1; echo "rc=$?" >> $O/t3_fixup.log
  tail -2 $O/t3_fixup.log | tee -a $O/summary.txt
  timeout 200 python scripts/latency_mode.py 120 > $O/lat_default.txt 2>&1
  Y7T_LIB=$LIBD/exp_fixup.so Y7T_CONV_SPLITK=2 timeout 200 python scripts/latency_mode.py 120 > $O/lat_fixup.txt 2>&1
  for f in default fixup; do echo "-- $f"; grep -i "fps" $O/lat_$f.txt | tail -8; done | tee -a $O/summary.txt
else say "exp_fixup.so missing"; fi
fi

# ---- 4. next: tracker candidate lists on a run-time row stride (LDS-resident at 500 objects) ----
if want 4; then
say "4. next (Y7T_NEXT_TRACKER): tracker parity on the device, then the 500-object frame step against the default library"
if [ -f $LIBD/exp_next.so ]; then
  Y7T_LIB=$LIBD/exp_next.so timeout 250 python -m pytest tests/test_tracker_gpu.py -q -m gpu > $O/t4_next.log 2>&1; echo "rc=$?" >> $O/t4_next.log
  tail -2 $O/t4_next.log | tee -a $O/summary.txt
  timeout 150 python scripts/time_tracker.py > $O/trk_default.txt 2>&1
  Y7T_LIB=$LIBD/exp_next.so timeout 150 python scripts/time_tracker.py > $O/trk_next.txt 2>&1
  for f in default next; do echo "-- $f"; grep -v amdgpu.ids $O/trk_$f.txt | tail -12; done | tee -a $O/summary.txt
else say "exp_next.so missing"; fi
fi

# ---- 5. stride-1 LDS-patch kernel with the step's DMAs behind its MFMAs (ABL bit 9: k_conv3x3_patch<..., 512>; correct results) ----
if want 5; then
say "5. patch kernel, DMA-late order (Y7T_CONV_ABLATE=512): parity in the benchmarked list, then the 3x3 / stride-1 rows against 1c's default table"
Y7T_CONV_ABLATE=512 timeout 250 python -m pytest tests/test_detector_pinned_gpu.py -q -m gpu -k "every_op" > $O/t5_late_pinned.log 2>&1; echo "rc=$?" >> $O/t5_late_pinned.log
tail -2 $O/t5_late_pinned.log | tee -a $O/summary.txt
[ -f $O/b1c_default.txt ] || timeout 200 python scripts/bench_conv.py 32 > $O/b1c_default.txt 2>&1
Y7T_CONV_ABLATE=512 timeout 200 python scripts/bench_conv.py 32 > $O/b5_late.txt 2>&1
for f in b1c_default b5_late; do echo "-- $f (the 64-channel layers run single-tile under the switch: compare the 128-channel and 40x40 rows)"; grep " 3/1 \|TOTAL" $O/$f.txt; done | tee -a $O/summary.txt
fi

say "done"
