#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2q; mkdir -p $O
export TMPDIR=/tmp
for v in 4 7 5 6 8; do
  Y7T_CONV_WPANEL=0 Y7T_CONV_VARIANT=$v timeout 300 python scripts/bench_conv.py 32 10 > $O/var$v.txt 2>&1
done
tail -n 1 $O/var*.txt
