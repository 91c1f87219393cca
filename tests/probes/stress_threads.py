"""Several handles driven concurrently by host threads (the ABI promises safety across handles): results must match the oracle."""
import sys, os, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver
SHAPES = [(256, 32, 4), (900, 300, 20), (4096, 512, 64), (700, 300, 100), (1300, 600, 8), (512, 64, 8)]
probs = {sh: synth.make_problem(99000 + i, *sh) for i, sh in enumerate(SHAPES)}
refs = {sh: go.gn_subproblem(*probs[sh]).p for sh in SHAPES}
errs = []
def worker(tid):
    rng = np.random.default_rng(tid)
    s = GNSolver(device=0)
    for k in range(60):
        sh = SHAPES[int(rng.integers(0, len(SHAPES)))]
        out = s.solve(*probs[sh])
        e = np.linalg.norm(out.p - refs[sh]) / np.linalg.norm(refs[sh])
        if not e <= 1e-10: errs.append((tid, sh, e))
    s.close()
ths = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]
print("errors", errs[:5], len(errs)); sys.exit(1 if errs else 0)
