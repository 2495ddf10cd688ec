#!/bin/bash
# MFMA-busy PMC pass over the conv kernels of ONE forward of the default bench workload -> gpurun_out/prof/conv_mfma_busy.json
# (own rocprofv3 run: --pmc with --kernel-trace only, wrapped in timeout)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/mb
CTRS="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
timeout 250 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/mb -- python $root/bench.py --steps 1 --warmup 1 --no_cpu_baseline > /tmp/mb.log 2>&1
python3 - $out "$CTRS" <<'PY'
import csv, glob, json, sys, collections
out, ctrs = sys.argv[1], sys.argv[2].split()
f = glob.glob("/tmp/mb/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
rows = [r for r in csv.DictReader(open(f[0])) if "k_conv" in r["Kernel_Name"]]
disp = collections.OrderedDict()
for r in rows:
    disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = list(disp)
half = ids[-(len(ids) // 3):]                    # three forwards ran (objectness planting at 1 frame, warm-up, timed): the last one
s = {c: sum(disp[i].get(c, 0.0) for i in half) for c in ctrs}
frames, gflop = 32, 354.9
exp = frames * gflop * 1e9 / (2.0 * 32 * 32 * 16)
cyc = s["GRBM_GUI_ACTIVE"] / 8.0                 # summed over the 8 XCDs
res = {"source": "scripts/pmc_mfma_busy.sh: rocprofv3 --kernel-trace --pmc " + " ".join(ctrs) + " -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline",
       "scope": "the %d k_conv* launches of the last forward (32 frames)" % len(half), "sum": s, "gpu_cycles_per_xcd": cyc,
       "mfma_busy_fraction": s["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), "mfma_insts": s["SQ_INSTS_MFMA"], "expected_mfma_insts": exp,
       "valu_per_mfma": s["SQ_INSTS_VALU"] / max(1.0, s["SQ_INSTS_MFMA"]),
       "wave_cycle_split": {"wait_any": s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], "wait_inst_any": s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"]}}
json.dump(res, open(out + "/conv_mfma_busy.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in ("mfma_busy_fraction", "mfma_insts", "expected_mfma_insts", "valu_per_mfma", "wave_cycle_split")}))
PY
