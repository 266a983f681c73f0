import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_step<" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-16:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(r.get("Queue_Id"), r.get("Stream_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X"))
