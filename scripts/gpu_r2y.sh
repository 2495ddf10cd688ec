#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2y; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python scripts/bench_conv.py 32 10 > $O/base.txt 2>&1
for v in a1 a2 w2 a2w2; do
  Y7T_LIB=$PWD/yolov7-tracker_amd/lib/aux_$v.so timeout 200 python scripts/bench_conv.py 32 10 > $O/$v.txt 2>&1
done
tail -n 1 $O/*.txt
