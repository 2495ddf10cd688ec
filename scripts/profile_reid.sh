#!/bin/bash
# PMC counters of the fused OSNet kernel (run on the GPU box from the repo root): LDS bank conflicts and MFMA share.  --pmc passes carry --kernel-trace only.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_reid; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/r1 /tmp/r2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d /tmp/r1 -- python $root/scripts/time_reid.py > $out/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/r2 -- python $root/scripts/time_reid.py > $out/p2.log 2>&1
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for d in ("/tmp/r1", "/tmp/r2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    acc = collections.defaultdict(float); n = 0
    grid = None
    for r in csv.DictReader(open(f[0])):
        if "k_osnet_x025" in r["Kernel_Name"] and int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0) >= 2560 * 512:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] == list(acc)[0]: n += 1
    for k, v in acc.items(): res[k] = v / max(n, 1)
    res["dispatches_" + d[-2:]] = n
if res:
    if "SQ_LDS_IDX_ACTIVE" in res: res["lds_bank_conflict_fraction"] = res["SQ_LDS_BANK_CONFLICT"] / max(res["SQ_LDS_IDX_ACTIVE"], 1)
    if "SQ_BUSY_CU_CYCLES" in res: res["mfma_busy_over_cu_busy"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / max(res["SQ_BUSY_CU_CYCLES"], 1)
    res["note"] = "per dispatch of k_osnet_x025 with 2560 crops (scripts/time_reid.py under rocprofv3 --kernel-trace --pmc ..., two separate passes)"
    json.dump(res, open(out + "/reid_pmc.json", "w"), indent=1)
    print(json.dumps(res, indent=1))
PY
