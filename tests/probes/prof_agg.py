"""Aggregate a rocprofv3 kernel trace CSV by kernel for launches whose grid contains `batch` problems."""
import csv, glob, sys
d, batch, steps = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
agg = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gn::" not in n:
        continue
    gx, gy, gz = int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
    nb = (gx // int(r["Workgroup_Size_X"]), gy, gz)
    if batch not in nb:
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    k = n.split("(")[0][-40:]
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += dur
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:42s} calls/step {v[0]/steps:6.1f} us/step {v[1]/steps:9.1f} avg {v[1]/v[0]:8.1f} us {100*v[1]/tot:5.1f}%")
print("total per step us", tot / steps)
