
"""reduce the rocprofv3 passes of `scripts/gpu_round.sh profile` (kernel trace of the launch list, FETCH_SIZE / WRITE_SIZE passes, MFMA-busy pass) into the
files the documents cite: conv_hbm_traffic.json (stamped with the commit and the launch list it was measured on -- bench.py prints roofline.traffic only
for that list), conv_mfma_busy.json, conv_per_layer_b32.txt.   python scripts/profile_reduce.py OUTDIR "CTR1 CTR2 ..." COMMIT"""
import csv, glob, json, sys, collections, re
out, ctrs, commit = sys.argv[1], sys.argv[2].split(), (sys.argv[3] if len(sys.argv) > 3 else "unknown")
try:
    META = json.loads([l for l in open(out + "/forward_only.log") if l.startswith("{")][-1])
except Exception:
    META = {}
FWD = ("k_conv", "k_stem", "k_splitk", "k_maxpool", "k_upsample", "k_spp3")
def last_forward(d, n_fwd):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: return None
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if any(k in r["Kernel_Name"] for k in FWD):
            disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    per = None
    # the forwards are the trailing n_fwd identical runs: find the period
    names = [disp[i]["name"] for i in ids]
    for L in range(60, 400):
        if len(names) >= 2 * L and names[-L:] == names[-2 * L:-L]:
            per = L; break
    if per is None: return None
    return [disp[i] for i in ids[-per:]]
res = {}
pf, pw = (last_forward("/tmp/pf", 3), last_forward("/tmp/pw", 3)) if ctrs else (None, None)      # (no counter list: the per-op table only -- scripts/per_layer_table.sh; a box may still hold an earlier call's /tmp)
if pf and pw:
    fetch_kb, write_kb = sum(d.get("FETCH_SIZE", 0) for d in pf), sum(d.get("WRITE_SIZE", 0) for d in pw)
    frames = int(META.get("B", 32))
    hbm = (2 * fetch_kb + write_kb) * 1024 / frames
    json.dump({"commit": commit, "launch_list_sha": META.get("launch_list_sha"), "weights": META.get("weights"),
               "source": "scripts/gpu_round.sh profile: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python scripts/forward_only.py 3",
               "kernels": "every launch of one forward of the benchmarked launch list (fused stem, convs, split-K reduces, pools; %d launches), %d frames" % (len(pf), frames),
               "frames_per_launch_list": frames, "FETCH_SIZE_KB_per_launch_list": fetch_kb, "WRITE_SIZE_KB_per_launch_list": write_kb,
               "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE as reported (uncalibrated)",
               "hbm_bytes_per_frame": hbm, "algorithmic_bytes_per_frame": 1217000000.0, "ratio_to_algorithmic": hbm / 1217000000.0},
              open(out + "/conv_hbm_traffic.json", "w"), indent=1)
    print(open(out + "/conv_hbm_traffic.json").read())
mb = last_forward("/tmp/mb", 3) if ctrs else None
if mb:
    s = {c: sum(d.get(c, 0.0) for d in mb) for c in ctrs}
    frames, gflop = int(META.get("B", 32)), 354.9
    exp = frames * gflop * 1e9 / (2.0 * 32 * 32 * 16)
    cyc = s["GRBM_GUI_ACTIVE"] / 8.0
    r = {"commit": commit, "launch_list_sha": META.get("launch_list_sha"), "source": "scripts/gpu_round.sh profile: rocprofv3 --kernel-trace --pmc " + " ".join(ctrs) + " -- python scripts/forward_only.py 3",
         "scope": "the %d launches of the last forward (%d frames)" % (len(mb), frames), "sum": s, "gpu_cycles_per_xcd": cyc,
         "mfma_busy_fraction": s["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), "mfma_insts": s["SQ_INSTS_MFMA"], "expected_mfma_insts": exp,
         "valu_per_mfma": s["SQ_INSTS_VALU"] / max(1.0, s["SQ_INSTS_MFMA"]),
         "wave_cycle_split": {"wait_any": s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], "wait_inst_any": s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"]}}
    json.dump(r, open(out + "/conv_mfma_busy.json", "w"), indent=1)
    print(json.dumps({k: r[k] for k in ("mfma_busy_fraction", "mfma_insts", "expected_mfma_insts", "valu_per_mfma", "wave_cycle_split")}))
# per-op table from the kernel trace
try:
    meta = json.loads([l for l in open(out + "/forward_only.log") if l.startswith("{")][-1])
    rows = [r for r in csv.DictReader(open(out + "/forward_kernel_trace.csv")) if any(k in r["Kernel_Name"] for k in FWD)]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))      # (the trace file is not in launch order when kernels last a few microseconds: batch 1)
    names = [r["Kernel_Name"] for r in rows]
    per = next(L for L in range(60, 400) if len(names) >= 2 * L and names[-L:] == names[-2 * L:-L])      # (batch 1: ~190 launches with the split-K reduces)
    last = rows[-per:]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
    # map dispatches to ops: a split-K conv is followed by its reduce
    it = iter(zip(last, dur))
    lines, tot_us, tot_gf = [], 0.0, 0.0
    spp_skip = 0
    for op in meta["ops"]:
        if spp_skip:                      # the second and third pool of an SPP cascade ran inside the first one's launch (k_spp3_lds)
            spp_skip -= 1
            lines.append("%3d %-44s %4dx%-4d %4d->%-4d %d/%d %9s    (in the launch of the row above)" % (op["op"], op["kernel"], op["H"], op["W"], op["Cin"], op["Cout"], op["k"], op["s"], "-"))
            continue
        r, d = next(it)
        for _ in range(int(op.get("launches", 1)) - 1):      # a conv that went out as several launches over runs of frames (tensors above 2 GiB)
            d += next(it)[1]
        if "k_spp3" in r["Kernel_Name"]:
            spp_skip = 2
            op = dict(op, kernel="spp3<5,5,5> lds")
        if "splitK" in op["kernel"]:
            r2, d2 = next(it); d += d2
        gf, by = op.get("gflop", 0.0), op.get("bytes", 0)
        lines.append("%3d %-44s %4dx%-4d %4d->%-4d %d/%d %9.1f us %8.1f TF/s %7.0f GB/s" % (op["op"], op["kernel"], op["H"], op["W"], op["Cin"], op["Cout"], op["k"], op["s"],
                                                                                    d, gf / d * 1e3 if d else 0, by / d * 1e-3 if d else 0))
        tot_us += d; tot_gf += gf
    nb = int(meta.get("B", 32))
    open(out + "/conv_per_layer_b%d.txt" % nb, "w").write("op kernel shape(HxW Cin->Cout k/s)  time  TFLOP/s  algorithmic GB/s   (one forward of %d frames, kernels timed back to back by rocprofv3 --kernel-trace)\n" % nb +
                                                   "\n".join(lines) + "\nTOTAL %.3f ms per %d frames -> %.1f TFLOP/s\n" % (tot_us / 1e3, nb, tot_gf / tot_us * 1e3))
    print("per-layer table: TOTAL %.3f ms -> %.1f TFLOP/s" % (tot_us / 1e3, tot_gf / tot_us * 1e3))
except Exception as e:
    print("per-layer table failed:", repr(e))
    try:      # what did not line up: dispatches of the last forward against the ops the launch list expects
        print("  dispatches in the trace: %d, period %s; ops %d, of them flagged splitK %d" % (len(names), locals().get("per"), len(meta["ops"]), sum("splitK" in o["kernel"] for o in meta["ops"])))
        short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:60]
        tail = [short(n) for n in names[-(locals().get("per") or 220):]]
        print("  last dispatches:", " | ".join(tail))
        print("  expected:", " | ".join(o["kernel"][:40] for o in meta["ops"]))
    except Exception as e2:
        print("  (no diagnostics:", repr(e2), ")")
