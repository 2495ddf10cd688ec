#!/bin/bash
# Diagnosis (step `pmc_queues` of scripts/gpu_round.sh): WHAT bounds the buffer->LDS path of the generic implicit-GEMM kernel?  Round 2 established that its 1x1 and stride-2 layers take
# ~1 us per K-step and workgroup whatever the K-step carries (ring depth, BK, tile and weight layout do not move it) but not which queue that
# microsecond is spent in.  The counters below separate the candidates:
#   issue side      SQ_INST_CYCLES_VMEM_RD / SQ_INSTS_VMEM_RD (cycles to send one wave's addresses), SQ_VMEM_TA_{ADDR,CMD}_FIFO_FULL, SQ_ACTIVE_INST_VMEM
#   address unit    TA_BUSY, TA_BUFFER_TOTAL_CYCLES / TA_BUFFER_READ_LDS_WAVEFRONTS (cycles per wave-DMA in the TA), TA_*_STALLED_BY_{TC,TD}
#   vector L1       TCP_PENDING_STALL (waiting for L2), TCP_{L,R}FIFO_STALL, TCP_TCR_RDRET_STALL, tag conflicts, requests to L2 per tag access, READ_REQ_LATENCY / READ_REQ
#   L2 / fabric     TCC_HIT / MISS / TAG_STALL, TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ (reads in flight towards memory)
#   LDS side        SQ_LDS_{DATA,CMD}_FIFO_FULL, SQ_LDS_BANK_CONFLICT
# on three layers at 32 frames: the 1x1 80x80 1024->512 (generic, 1x1 weight panels), the stride-2 3x3 160x160 256->512 (generic), and for contrast the
# stride-1 3x3 80x80 256->256 on the LDS-patch kernel (1000+ TFLOP/s).  Each counter group is its own rocprofv3 run (--pmc with --kernel-trace only).
#   OUT=<dir> bash scripts/pmc_queues.sh                      (about 36 short runs; SHAPES="1x1" or "s2" or "patch" restricts)
O=${OUT:-$GRAFT_REPO_ROOT/gpurun_out/pmc_queues}
GROUPS_=(
 "TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
 "TCP_TCR_RDRET_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum"
 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum"
 "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVE_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
)
run_shape() {   # name, "H W Cin Cout k s B", ACT_BITS
  mkdir -p $O/$1
  ACT_BITS=$3 bash $GRAFT_REPO_ROOT/scripts/pmc_conv.sh $O/$1 "$2" "${GROUPS_[@]}" > $O/$1/log.txt 2>&1
  echo "=== $1 ($2, act bits $3)"; grep -c mean $O/$1/summary.txt
}
S=${SHAPES:-"1x1 s2 patch"}
for s in $S; do
  case $s in
    1x1)   run_shape conv1x1_80_1024_512 "80 80 1024 512 1 1 32" 2048 ;;
    s2)    run_shape conv3x3s2_160_256_512 "160 160 256 512 3 2 32" 256 ;;
    patch) run_shape patch3x3_80_256_256 "80 80 256 256 3 1 32" 1024 ;;
  esac
done
# one table: counter -> value per kernel, normalised per wave-DMA where that makes sense
python3 - <<'PY'
import collections, os, re
root = os.environ.get("OUT") or os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_queues")
for d in sorted(os.listdir(root)):
    f = os.path.join(root, d, "summary.txt")
    if not os.path.isfile(f):
        continue
    v = {}
    for line in open(f):
        m = re.match(r"(.{62}) (\S+)\s+n=(\d+) mean=(\S+)", line)
        if m:
            v[m.group(2)] = float(m.group(4))
    print("==", d)
    for k in sorted(v):
        print("  %-44s %.4g" % (k, v[k]))
    w = v.get("TA_BUFFER_READ_LDS_WAVEFRONTS_sum") or v.get("SQ_INSTS_VMEM_RD")
    if w:
        for k in ("TA_BUFFER_TOTAL_CYCLES_sum", "SQ_INST_CYCLES_VMEM_RD", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum"):
            if k in v:
                print("  per wave-DMA: %-30s %.2f" % (k, v[k] / w))
    if v.get("TCP_TCC_READ_REQ_sum") and v.get("TCP_TCC_READ_REQ_LATENCY_sum"):
        print("  mean L1->L2 read latency (cycles): %.0f" % (v["TCP_TCC_READ_REQ_LATENCY_sum"] / v["TCP_TCC_READ_REQ_sum"]))
    if v.get("TCC_EA0_RDREQ_sum") and v.get("TCC_EA0_RDREQ_LEVEL_sum"):
        print("  mean L2->memory read latency (cycles): %.0f" % (v["TCC_EA0_RDREQ_LEVEL_sum"] / v["TCC_EA0_RDREQ_sum"]))
PY
