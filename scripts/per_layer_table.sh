#!/bin/bash
# per-op table of the benchmarked launch list (one forward of bench.DEFAULT_BATCH frames -- or B=<n> --, kernels back to back, rocprofv3 --kernel-trace) under the CURRENT environment:
#   NAME=<label> OUT=<dir> bash scripts/per_layer_table.sh      -> $OUT/per_layer_$NAME.txt
# (the table part of `scripts/gpu_round.sh profile`; A/B experiments call it once per variant inside one session)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${OUT:-$ROOT/gpurun_out/per_layer}; NAME=${NAME:-default}
T=/tmp/plt_$NAME; rm -rf $T /tmp/kt_$NAME; mkdir -p $T $OUT
( cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$NAME -- python $ROOT/scripts/forward_only.py 4 > $T/forward_only.log 2>&1
  f=$(find /tmp/kt_$NAME -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $T/forward_kernel_trace.csv
  python3 $ROOT/scripts/profile_reduce.py $T "" unknown > $T/reduce.log 2>&1; grep "per-layer" $T/reduce.log )
cp $(ls $T/conv_per_layer_b*.txt 2>/dev/null | head -1) $OUT/per_layer_$NAME.txt 2>/dev/null || { echo "no table for $NAME"; cat $T/reduce.log | cut -c1-12000; tail -5 $T/forward_only.log | cut -c1-300; }
