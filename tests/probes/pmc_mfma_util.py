"""f64 matrix-pipe utilisation of the two MFMA kernels from two rocprofv3 PMC passes (evidence tooling, not a test).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d out_b -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-roofline
    rocprofv3 --pmc GRBM_GUI_ACTIVE          --output-format csv -d out_g -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-roofline
    python3 tests/probes/pmc_mfma_util.py <busy counter_collection.csv> <gui counter_collection.csv> [out.json]

One counter per pass.  For each kernel only its largest-grid launches are taken (the panel-0 update / the J*Q1 of a pipelined
half batch).  GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel cycles = value / 8); SQ_VALU_MFMA_BUSY_CYCLES counts 64 cycles
per v_mfma_f64_16x16x4_f64 and SIMD (calibrated on k_jq1_v2's known MFMA count, profiles/r1_notes.md);
utilisation = busy cycles / (kernel cycles x 1024 SIMDs)."""
import csv, json, sys

KERNELS = {"k_caqr_update_v4_pair<8>": "k_caqr_update_v4_pair<8> (level-0 far update, two panels per pass)",
           "k_caqr_update_v4<8, false>": "k_caqr_update_v4<8,false> (level-0 update of one panel: the pair's second panel's columns)",
           "k_jq1_v2<8, 4, 2>": "k_jq1_v2<8,4,2> (J*Q1)"}
SIMDS = 1024


def largest(path, counter, key):
    rows = [(int(r["Grid_Size"]), float(r["Counter_Value"])) for r in csv.DictReader(open(path))
            if key in r["Kernel_Name"] and r["Counter_Name"] == counter]
    g = max(x for x, _ in rows)
    v = [c for x, c in rows if x == g]
    return {"launches": len(v), "grid_threads": g, "per_launch": sum(v) / len(v)}


def main():
    busy_csv, gui_csv = sys.argv[1], sys.argv[2]
    out = {"kernels": {}}
    for key, label in KERNELS.items():
        if not any(key in r["Kernel_Name"] for r in csv.DictReader(open(busy_csv))):
            continue
        b = largest(busy_csv, "SQ_VALU_MFMA_BUSY_CYCLES", key)
        g = largest(gui_csv, "GRBM_GUI_ACTIVE", key)
        cyc = g["per_launch"] / 8.0
        out["kernels"][label] = {"SQ_VALU_MFMA_BUSY_CYCLES": b, "GRBM_GUI_ACTIVE": g, "kernel_cycles": cyc,
                                 "mfma_utilisation": round(b["per_launch"] / (cyc * SIMDS), 4)}
        print(label, out["kernels"][label]["mfma_utilisation"])
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
