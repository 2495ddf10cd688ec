#!/bin/bash
# PMC passes over ONE conv shape (rocprofv3 --pmc, each pass in its own run, wrapped in timeout):
#   scripts/pmc_conv.sh OUTDIR "H W Cin Cout k s B" "CTR1 CTR2 ..." ["CTR..." ...]
out=$1; shape=$2; shift 2
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$i -- python $GRAFT_REPO_ROOT/scripts/sweep_conv.py $shape 2 > /tmp/pmc_$i.log 2>&1 || { echo "pass $i failed/timeout"; tail -5 /tmp/pmc_$i.log; }
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" >> "$out/summary.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "conv" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-62s %-36s n=%d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
PY
done
cat "$out/summary.txt"
