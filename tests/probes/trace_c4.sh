# kernel timeline of C4's local stage (tests/probes/trace_c4.py); GPU box
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/c4trace}; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --config C4 --steps 2 --warmup 1 --cpu-budget 0 --no-roofline > $O/bench.json 2> $O/kt.err
python3 tests/probes/trace_c4.py $O/kt > $O/summary.txt
cat $O/summary.txt
python3 - <<PY
import csv, glob
f = glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gn::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
packs = [i for i, r in enumerate(rows) if "k_tsqr_pack" in r["Kernel_Name"]]
hi = packs[-1]
# one pair in the middle of the last solve: print ~60 launches around the 8th pair update from the end
pairs = [i for i in range(hi) if "update_v4_pair" in rows[i]["Kernel_Name"]]
mid = pairs[-14]
t0 = int(rows[mid]["Start_Timestamp"])
with open("$O/one_pair.txt", "w") as o:
    for r in rows[mid:mid + 64]:
        k = r["Kernel_Name"].split("(")[0].split("gn::")[-1][:34]
        o.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  q{r.get('Queue_Id', '?')} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}  {k}\n")
print(open("$O/one_pair.txt").read())
PY
rm -rf $O/kt
