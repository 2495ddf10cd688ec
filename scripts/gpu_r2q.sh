#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2q2; mkdir -p $O
export TMPDIR=/tmp
for v in 5 6 8; do
  Y7T_CONV_WPANEL=0 Y7T_CONV_VARIANT=$v Y7T_LIB=$PWD/yolov7-tracker_amd/lib/ablate_4.so timeout 300 python scripts/bench_conv.py 32 10 > $O/nomfma_var$v.txt 2>&1
  Y7T_CONV_WPANEL=0 Y7T_CONV_VARIANT=$v timeout 300 python scripts/bench_conv.py 32 10 > $O/full_var$v.txt 2>&1
done
tail -n 1 $O/*.txt
