"""Timeline of the two pipelined halves of a default bench.py step from a rocprofv3 kernel trace (diagnostic tooling, not a test).

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-roofline
    python3 tests/probes/trace_overlap.py out

For the launches of the timed steps (grids of batch/2 problems) it prints, per kernel class, its busy time per step, the time during
which a kernel of the OTHER queue ran concurrently, and the union / idle time of the whole step: how much of the latency-bound work
(panel factorisations, pivot steps) actually hides behind the bandwidth-bound kernels of the other half."""
import csv, glob, sys
from collections import defaultdict

d = sys.argv[1]
half = int(sys.argv[2]) if len(sys.argv) > 2 else 192
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gn::" not in n:
        continue
    gx, gy, gz = int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
    nb = (gx // int(r["Workgroup_Size_X"]), gy, gz)
    if half not in nb:
        continue
    q = r.get("Queue_Id", r.get("Stream_Id", "0"))
    k = n.split("(")[0].split("gn::")[-1][:34]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), q, k))
rows.sort()
queues = sorted({r[2] for r in rows})
print("queues:", queues, "launches:", len(rows))
t0, t1 = rows[0][0], max(r[1] for r in rows)
# busy intervals per queue
def merge(iv):
    iv = sorted(iv); out = []
    for a, b in iv:
        if out and a <= out[-1][1]: out[-1][1] = max(out[-1][1], b)
        else: out.append([a, b])
    return out
busy = {q: merge([(a, b) for a, b, qq, _ in rows if qq == q]) for q in queues}
def overlap(a, b, iv):
    return sum(max(0, min(b, y) - max(a, x)) for x, y in iv)
tot = defaultdict(float); ovl = defaultdict(float); cnt = defaultdict(int)
for a, b, q, k in rows:
    others = [iv for qq, ivs in busy.items() if qq != q for iv in ivs]
    tot[k] += b - a; cnt[k] += 1
    ovl[k] += overlap(a, b, others)
union = merge([(a, b) for a, b, _, _ in rows])
ub = sum(b - a for a, b in union)
print(f"span {1e-6 * (t1 - t0):.2f} ms, union busy {1e-6 * ub:.2f} ms, sum of kernel times {1e-6 * sum(tot.values()):.2f} ms")
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f"{k:36s} n {cnt[k]:5d}  busy {1e-6 * tot[k]:8.2f} ms  with another queue's kernel running {100 * ovl[k] / tot[k]:5.1f} %  avg {1e-3 * tot[k] / cnt[k]:8.1f} us")
