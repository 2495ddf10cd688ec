#!/bin/bash
# SQ / LDS / VMEM issue counters of the kernels whose name matches a pattern, inside any command (rocprofv3 --pmc, one counter group per run, --kernel-trace only):
#   OUT=<dir> bash scripts/pmc_kernel.sh '<name substring>' python scripts/forward_only.py 2
pat=$1; shift
O=${OUT:-$GRAFT_REPO_ROOT/gpurun_out/pmc_kernel}; mkdir -p $O; : > $O/summary.txt
cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" \
  "SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pk_$i
  ( cd $GRAFT_REPO_ROOT && timeout 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pk_$i -- "$@" > /tmp/pk_$i.log 2>&1 ) || { echo "pass $i failed"; tail -3 /tmp/pk_$i.log; }
  f=$(find /tmp/pk_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$pat" >> $O/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-50s %-30s n=%d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
PY
done
cat $O/summary.txt
