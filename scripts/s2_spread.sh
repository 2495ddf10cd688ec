#!/bin/bash
# The stride-2 patch kernel with its fragment reads spread over the step's MFMAs (Y7T_CONV_PATCH_S2_ORDER=2) against the default order, same session:
# parity of the layer cases and of the benchmarked list (teacher-forced) with it, then per-layer time of both.     OUT=<dir> bash scripts/s2_spread.sh
O=${OUT:-$GRAFT_REPO_ROOT/gpurun_out/s2_spread}; mkdir -p $O; cd $GRAFT_REPO_ROOT
Y7T_CONV_PATCH_S2_ORDER=2 timeout 100 python -m pytest tests/test_detector_gpu.py tests/test_detector_pinned_gpu.py -q -m gpu -k "stride2 or every_op" > $O/t_s2.log 2>&1; tail -1 $O/t_s2.log
for shape in "320 320 128 256" "160 160 256 512" "80 80 512 768"; do
  for o in 0 2; do
    r=$(Y7T_CONV_PATCH_S2_ORDER=$o ACT_BITS=4096 timeout 40 python scripts/sweep_conv.py $shape 3 2 32,32 20 2>/dev/null | tail -1)
    echo "$shape 3/2  order $o: $r"
  done
done | tee $O/s2_spread.txt
