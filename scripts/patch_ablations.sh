#!/bin/bash
# Timing ablations (wrong results) of the LDS-patch 3x3 kernel in its current form, two layers of the benchmarked list at 32 frames:
#   Y7T_CONV_ABLATE bits: 1 zero-filling DMAs only (no real memory traffic), 2 no MFMAs, 4 no fragment reads, 8 no epilogue, 16 no weight-panel traffic, 32 no patch pieces
#   OUT=<dir> bash scripts/patch_ablations.sh
export Y7T_LIB=${Y7T_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/yolov7-tracker_amd/lib/liby7t_ablate.so}      # experiment switches / ablation instances live in the measuring build
O=${OUT:-$GRAFT_REPO_ROOT/gpurun_out/patch_ablations}; mkdir -p $O; cd $GRAFT_REPO_ROOT
for shape in "160 160 128 128" "80 80 256 256"; do
  echo "== $shape 3/1, B = 32 (second of two timings per process)"
  for a in 0 8 4 2 16 32 48 1 15; do
    r=$(Y7T_CONV_ABLATE=$a ACT_BITS=1024 timeout 60 python scripts/sweep_conv.py $shape 3 1 32,32 30 2>/dev/null | tail -1)
    echo "ablate $a: $r"
  done
done | tee $O/patch_ablations.txt
