#!/bin/bash
# Round profile of the default bench workload (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats of `python bench.py`            -> gpurun_out/prof/kernel_stats.csv
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) at 8 frames    -> gpurun_out/prof/conv_hbm_traffic.json
# Every rocprofv3 run is wrapped in `timeout`; PMC passes never carry trace flags other than --kernel-trace.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks /tmp/pf /tmp/pw
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $root/bench.py --no_cpu_baseline > $out/bench_under_rocprof.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python $root/bench.py --steps 2 --warmup 1 --batch 8 --no_cpu_baseline > /tmp/pf.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python $root/bench.py --steps 2 --warmup 1 --batch 8 --no_cpu_baseline > /tmp/pw.log 2>&1
python3 - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def per_list(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: return None, 0
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f[0])):
        if "k_conv" in r["Kernel_Name"] and r["Counter_Name"] == name:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n
fs, nf = per_list("/tmp/pf", "FETCH_SIZE")
ws, nw = per_list("/tmp/pw", "WRITE_SIZE")
if fs is None or ws is None:
    print("PMC pass missing"); sys.exit(0)
forwards = 3            # warmup 1 + steps 2
frames = 8
fetch_kb, write_kb = fs / forwards, ws / forwards
hbm = (2 * fetch_kb + write_kb) * 1024 / frames
json.dump({"source": "scripts/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python bench.py --steps 2 --warmup 1 --batch 8 --no_cpu_baseline",
           "kernels": "every k_conv* launch of one forward (k_conv_igemm + k_conv3x3_patch + k_splitk_reduce), %d launches per forward" % (nf // forwards),
           "frames_per_launch_list": frames, "FETCH_SIZE_KB_per_launch_list": fetch_kb, "WRITE_SIZE_KB_per_launch_list": write_kb,
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE as reported",
           "hbm_bytes_per_frame": hbm, "algorithmic_bytes_per_frame": 1217000000.0, "ratio_to_algorithmic": hbm / 1217000000.0},
          open(out + "/conv_hbm_traffic.json", "w"), indent=1)
print(open(out + "/conv_hbm_traffic.json").read())
PY
head -12 $out/kernel_stats.csv 2>/dev/null
tail -n 1 $out/bench_under_rocprof.log | cut -c1-300
