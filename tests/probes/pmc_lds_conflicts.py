"""LDS bank-conflict fraction per kernel from two rocprofv3 PMC passes (evidence tooling, not a test).

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT --output-format csv -d out_c -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-roofline
    rocprofv3 --pmc SQ_LDS_IDX_ACTIVE   --output-format csv -d out_a -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-roofline
    python3 tests/probes/pmc_lds_conflicts.py <conflict csv> <active csv> [out.json]

Sums over the batch launches (grid > 10^5 threads) of each library kernel; conflict_fraction = SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE (cycles the LDS spent resolving bank conflicts over the cycles it was busy with indexed accesses)."""
import csv, json, sys


def agg(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if "gn::" not in r["Kernel_Name"] or r["Counter_Name"] != counter or int(r["Grid_Size"]) < 100000:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = out.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return out


def main():
    c = agg(sys.argv[1], "SQ_LDS_BANK_CONFLICT")
    a = agg(sys.argv[2], "SQ_LDS_IDX_ACTIVE")
    rec = {"command": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT (and, separately, --pmc SQ_LDS_IDX_ACTIVE) --output-format csv -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-roofline",
           "kernels": {}}
    for k in sorted(a, key=lambda k: -a[k][1]):
        if k in c and a[k][1] > 0:
            rec["kernels"][k] = {"launches": a[k][0], "SQ_LDS_IDX_ACTIVE": a[k][1], "SQ_LDS_BANK_CONFLICT": c[k][1],
                                 "conflict_fraction": round(c[k][1] / a[k][1], 4)}
            print(f"{k[:44]:44s} active {a[k][1]:.3e} conflict {c[k][1]:.3e} fraction {c[k][1] / a[k][1]:.4f}")
    if len(sys.argv) > 3:
        json.dump(rec, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
