#!/bin/bash
# The patch kernels' two read schedules, same session: fragment reads spread over the step's MFMAs (default) against one burst behind the barrier
# (Y7T_CONV_ABLATE=2048; the form of rounds 1-3a): parity of the conv layer cases with the latter, then per-layer time of both.     OUT=<dir> bash scripts/patch_spread.sh
export Y7T_LIB=${Y7T_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/yolov7-tracker_amd/lib/liby7t_ablate.so}      # experiment switches / ablation instances live in the measuring build
O=${OUT:-$GRAFT_REPO_ROOT/gpurun_out/patch_spread}; mkdir -p $O; cd $GRAFT_REPO_ROOT
Y7T_CONV_ABLATE=2048 timeout 200 python -m pytest tests/test_detector_gpu.py -q -m gpu -k conv_layer > $O/t_spread.log 2>&1; tail -1 $O/t_spread.log
for shape in "160 160 128 128" "80 80 256 256" "40 40 384 384" "160 160 128 64" "80 80 128 128" "160 160 128 256"; do
  for a in 2048 0; do
    r=$(Y7T_CONV_ABLATE=$a ACT_BITS=1024 timeout 60 python scripts/sweep_conv.py $shape 3 1 32,32 30 2>/dev/null | tail -1)
    echo "$shape  schedule $a: $r"
  done
done | tee $O/patch_spread.txt
