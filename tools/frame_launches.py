import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
n_frames = int(sys.argv[2])
# steady state: count kernels between consecutive k_livox_prep launches
idx = [i for i, r in enumerate(rows) if "k_livox_prep" in r["Kernel_Name"]]
per = [idx[k + 1] - idx[k] for k in range(len(idx) - 1)]
per = per[len(per) // 2:]
busy = []
for k in range(len(idx) // 2, len(idx) - 1):
    seg = rows[idx[k]:idx[k + 1]]
    busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3)
    span = (int(rows[idx[k + 1]]["Start_Timestamp"]) - int(rows[idx[k]]["Start_Timestamp"])) / 1e3
print("launches per frame (median)", sorted(per)[len(per) // 2], "GPU busy us per frame (median)", sorted(busy)[len(busy) // 2])
