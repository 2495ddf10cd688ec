"""rocprofv3 --kernel-trace csv -> compact rows `short kernel name, queue, start (us since the first kernel), duration (us)` in start order
(python scripts/trace_compact.py IN.csv OUT.csv): what the round's timeline questions need, small enough to travel back from the GPU box."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:48]
with open(sys.argv[2], "w") as f:
    for r in rows:
        f.write("%s,%s,%.1f,%.1f\n" % (short(r["Kernel_Name"]), r.get("Queue_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
