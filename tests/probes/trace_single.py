"""Kernel timeline of ONE C2 solve (batch 1) from a rocprofv3 kernel trace (diagnostic tooling, not a test):
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --config C2 --batch 1 --steps 6 --warmup 3 --cpu-budget 0 --no-roofline --no-live-pmc
    python3 tests/probes/trace_single.py out
Takes the last solve (the kernels after the last k_vt / k_constraint launch): per kernel class launches, busy time, and the idle
time in front of its launches (start - end of the previous kernel on any queue)."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gn::" not in n:
        continue
    k = n.split("(")[0].split("gn::")[-1]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), k, n))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[3].startswith("k_constraint")]
lo = starts[-1]
sel = rows[lo:]
tot = defaultdict(float); cnt = defaultdict(int); gap = defaultdict(float)
prev_end = sel[0][0]
for a, b, q, k, n in sel:
    tot[k] += b - a; cnt[k] += 1
    if a > prev_end: gap[k] += a - prev_end
    prev_end = max(prev_end, b)
span = max(r[1] for r in sel) - sel[0][0]
print(f"launches {len(sel)}; span {span * 1e-3:.1f} us; sum of kernel times {sum(tot.values()) * 1e-3:.1f} us; idle {sum(gap.values()) * 1e-3:.1f} us")
for k in sorted(tot, key=lambda k: -(tot[k] + gap[k])):
    print(f"{k[:44]:46s} n {cnt[k]:4d}  busy {tot[k] * 1e-3:8.1f} us  avg {tot[k] * 1e-3 / cnt[k]:7.1f} us  idle in front {gap[k] * 1e-3:7.1f} us")
if len(sys.argv) > 2:
    t0 = sel[0][0]
    with open(sys.argv[2], "w") as o:
        for a, b, q, k, n in sel:
            o.write(f"{(a - t0) * 1e-3:9.1f} {(b - a) * 1e-3:8.1f} us q{q} {n.split('gn::')[-1][:70]}\n")
