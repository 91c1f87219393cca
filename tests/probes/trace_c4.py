"""Kernel timeline of C4's local stage from a rocprofv3 kernel trace (diagnostic tooling, not a test):
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --config C4 --steps 2 --warmup 1 --cpu-budget 0 --no-roofline
    python3 tests/probes/trace_c4.py out
Per kernel class of the LAST solve: launches, busy time, share that ran beside a kernel of the other queue; span / union / idle."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gn::" not in n:
        continue
    q = r.get("Queue_Id", r.get("Stream_Id", "0"))
    k = n.split("(")[0].split("gn::")[-1][:40]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), q, k))
rows.sort()
# the last solve starts with the last k_constraint / first kernel after a long gap: take the last third of the trace by k_tsqr_pack markers
packs = [i for i, r in enumerate(rows) if "k_tsqr_pack" in r[3]]
lo = 0
if len(packs) >= 2:
    # local stage of the last solve = kernels between the end of the previous solve's last kernel and the last pack
    prev_end = packs[-2]
    # skip the previous solve's combine stage: find the first k_jq1 / k_constraint after prev_end
    for i in range(prev_end, packs[-1]):
        if "k_jq1" in rows[i][3] or "k_constraint" in rows[i][3]:
            lo = i
            break
sel = rows[lo:packs[-1]] if packs else rows
def merge(iv):
    iv = sorted(iv); out = []
    for a, b in iv:
        if out and a <= out[-1][1]: out[-1][1] = max(out[-1][1], b)
        else: out.append([a, b])
    return out
queues = sorted({r[2] for r in sel})
busy = {q: merge([(a, b) for a, b, qq, _ in sel if qq == q]) for q in queues}
def overlap(a, b, iv):
    return sum(max(0, min(b, y) - max(a, x)) for x, y in iv)
tot = defaultdict(float); ovl = defaultdict(float); cnt = defaultdict(int)
for a, b, q, k in sel:
    others = [iv for qq, ivs in busy.items() if qq != q for iv in ivs]
    tot[k] += b - a; cnt[k] += 1; ovl[k] += overlap(a, b, others)
t0, t1 = sel[0][0], max(r[1] for r in sel)
union = merge([(a, b) for a, b, _, _ in sel])
ub = sum(b - a for a, b in union)
print(f"queues {queues}; launches {len(sel)}; span {1e-6 * (t1 - t0):.2f} ms, union busy {1e-6 * ub:.2f} ms, sum of kernel times {1e-6 * sum(tot.values()):.2f} ms")
for q in queues:
    print(f"  queue {q}: busy {1e-6 * sum(b - a for a, b in busy[q]):.2f} ms in {sum(1 for r in sel if r[2] == q)} launches")
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f"{k:42s} n {cnt[k]:5d}  busy {1e-6 * tot[k]:8.2f} ms  beside the other queue {100 * ovl[k] / tot[k]:5.1f} %  avg {1e-3 * tot[k] / cnt[k]:8.1f} us")
