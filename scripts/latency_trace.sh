#!/bin/bash
# Where a batch-1 frame's 2.9 ms go: rocprofv3 --kernel-trace of scripts/latency_mode.py (uint8 host frames, detector + NMS as a hipGraph replay), reduced to
# per-frame kernel time, gaps between kernels, launches, and the kernels by name.     OUT=<dir> bash scripts/latency_trace.sh
O=${OUT:-$GRAFT_REPO_ROOT/gpurun_out/latency_trace}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/lt_raw
( cd $GRAFT_REPO_ROOT && ONLY=u8,1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lt_raw -- python scripts/latency_mode.py 40 > $O/latency_mode_under_rocprof.log 2>&1 )
kt=$(find /tmp/lt_raw -name "*kernel_trace.csv" | head -1); mc=$(find /tmp/lt_raw -name "*memory_copy_trace.csv" | head -1)
python3 - "$kt" "$mc" <<'PY' | tee $O/latency_trace.txt
import csv, sys, collections
rows = [(float(r["Start_Timestamp"]), float(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# frames = groups of kernels separated by a host-side gap (sync + next frame's copy): split where the gap between consecutive kernels exceeds 150 us
frames, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 150e3: frames.append(cur); cur = []
    cur.append(b)
frames.append(cur)
frames = [f for f in frames if len(f) > 100][-30:]          # the timed steady state: whole frames only
print("frames analysed: %d, kernels per frame: %s" % (len(frames), sorted(set(len(f) for f in frames))))
span = [f[-1][1] - f[0][0] for f in frames]; busy = [sum(e - s for s, e, _ in f) for f in frames]
gaps = [sum(max(0.0, b[0] - a[1]) for a, b in zip(f, f[1:])) for f in frames]
print("per frame: first kernel start -> last kernel end %.0f us; kernel time %.0f us; idle between kernels %.0f us (%.1f us per launch)"
      % (sum(span) / len(span) / 1e3, sum(busy) / len(busy) / 1e3, sum(gaps) / len(gaps) / 1e3, sum(gaps) / len(gaps) / 1e3 / (len(frames[0]) - 1)))
acc = collections.defaultdict(lambda: [0, 0.0])
for f in frames:
    for s, e, n in f: acc[n[:80]][0] += 1; acc[n[:80]][1] += e - s
print("%-82s %7s %9s" % ("kernel", "n/frame", "us/frame"))
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]: print("%-82s %7.1f %9.1f" % (n, c / len(frames), t / len(frames) / 1e3))
if len(sys.argv) > 2 and sys.argv[2]:
    try:
        cp = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r.get("Direction", "")) for r in csv.DictReader(open(sys.argv[2]))]
        big = [d for d, _ in cp if d > 50e3]
        if big: print("host->device frame copies: %d, mean %.0f us" % (len(big), sum(big) / len(big) / 1e3))
    except Exception as ex: print("copy trace:", ex)
PY
