"""Which kernels touch their spill slots INSIDE a loop?  (Static, no GPU.)  A spill that is written once and reloaded after a loop costs
nothing; a reload inside a step loop is a vector-memory round trip per step, and its wait (vmcnt) drains whatever else is in flight.
usage:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I include -o /tmp/gn.s enlsip.jl_amd/csrc/enlsip_gn.hip -w
        python3 tests/probes/isa_scratch_in_loops.py /tmp/gn.s
Prints, per kernel with scratch traffic: scratch loads / stores in total and those that lie inside some loop (a label that a later
branch jumps back to), with the innermost loop's extent in lines."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
funcs = re.findall(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, flags=re.S | re.M)
def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n
for name, body in funcs:
    if "scratch_" not in body:
        continue
    lines = body.split("\n")
    lab = {}
    for i, l in enumerate(lines):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m: lab[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            loops.append((lab[m.group(1)], i))
    ld = [i for i, l in enumerate(lines) if "scratch_load" in l]
    st = [i for i, l in enumerate(lines) if "scratch_store" in l]
    def inner(i):
        c = [(b - a, a, b) for a, b in loops if a <= i <= b]
        return min(c) if c else None
    ld_in = [(i, inner(i)) for i in ld if inner(i)]
    st_in = [(i, inner(i)) for i in st if inner(i)]
    d = demangle(name).split("(")[0]
    print(f"{d[:70]:72s} loads {len(ld):3d} (in loops {len(ld_in):3d})  stores {len(st):3d} (in loops {len(st_in):3d})"
          + (f"  smallest enclosing loop {min(x[1][0] for x in ld_in + st_in)} lines" if ld_in or st_in else ""))
