#!/bin/bash
# Shader clock during each kernel of a command: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / 8 / the dispatch's duration from the kernel trace.
#   OUT=<dir> bash scripts/clock_probe.sh python scripts/bench_conv.py 32 20
# rocprofv3 --pmc with --kernel-trace only (one counter: one pass).  Prints, per kernel name, dispatches, mean duration and mean MHz.
O=${OUT:-$GRAFT_REPO_ROOT/gpurun_out/clock_probe}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/cp_raw
( cd $GRAFT_REPO_ROOT && timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/cp_raw -- "$@" > /tmp/cp_raw.log 2>&1 ) || { echo "rocprofv3 failed"; tail -5 /tmp/cp_raw.log; }
cc=$(find /tmp/cp_raw -name "*counter_collection.csv" | head -1); kt=$(find /tmp/cp_raw -name "*kernel_trace.csv" | head -1)
python3 - "$cc" "$kt" <<'PY' | tee $O/clock_probe.txt
import csv, sys, collections
cyc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cyc[r["Dispatch_Id"]] = float(r["Counter_Value"])
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[2])):
    d = r.get("Dispatch_Id")
    if d not in cyc: continue
    dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))      # ns
    if dur <= 0: continue
    acc.setdefault(r["Kernel_Name"][:70], []).append((dur, cyc[d] / 8.0 / dur * 1e3))
print("%-72s %6s %10s %8s" % ("kernel", "n", "mean us", "MHz"))
for k, v in acc.items():
    print("%-72s %6d %10.1f %8.0f" % (k, len(v), sum(a for a, _ in v) / len(v) / 1e3, sum(b for _, b in v) / len(v)))
PY
