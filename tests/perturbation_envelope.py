"""How far the line-search step lengths of an outer run move when the subproblem backend's arithmetic is perturbed at
rounding level — the yardstick for "iteration for iteration" comparisons of two backends (tests/test_reference_problems.py).

A backend that sums in another order than LAPACK (the HIP kernels, another OpenBLAS build, another thread count) returns
search directions that differ by cond x eps.  Most iterations do not care; a line search in a flat end game amplifies the
difference into its step length.  Instead of guessing a bound, the envelope is MEASURED on the oracle itself: the same outer
run is repeated with the Jacobian and the residuals handed to the LAPACK subproblem perturbed by one unit in the last place
(relative 2^-52, random signs, a different draw per call), and the per-iteration spread of the step length against the
unperturbed run is recorded.  Discrete outcomes (codes, ranks, working-set sizes, iteration count) are expected to be the same;
`agree_upto` says up to which iteration they were.

Test infrastructure: nothing here is on the product path."""
import numpy as np

from oracle import enlsip_outer as eo

ULP = 2.0 ** -52


class UlpPerturbedBackend(eo.OracleBackend):
    """The LAPACK oracle on inputs perturbed by +-1 ulp (relative), a fresh draw for every subproblem."""

    def __init__(self, seed):
        super().__init__()
        self.rng = np.random.default_rng(seed)

    def _jiggle(self, a):
        a = np.asarray(a, dtype=float)
        return a * (1.0 + ULP * self.rng.integers(-1, 2, size=a.shape))

    def update_working_set(self, W, rx, A, C, grad_fx, J, p_gn, it, eps_rank, on_solve):
        return super().update_working_set(W, self._jiggle(rx), A, C, grad_fx, self._jiggle(J), p_gn, it, eps_rank, on_solve)

    def sub_search_direction(self, J1, rx, cx, F_A, F_L11, F_J2, n, t, rankA, dimA, dimJ2, code):
        return super().sub_search_direction(self._jiggle(J1), self._jiggle(rx), cx, F_A, F_L11, F_J2, n, t, rankA, dimA, dimJ2, code)


def discrete(rec):
    return (rec["code"], rec["t"], rec["rankA"], rec["rankJ2"])


def alpha_envelope(run, ref, seeds=(1, 2, 3, 4, 5, 6)):
    """run(backend) -> result with .trace; ref = the unperturbed result.  Returns (env, agree_upto): env[i] = largest
    |alpha_i(perturbed) - alpha_i(ref)| / max(1, |alpha_i(ref)|) over the perturbed runs (iterations on which a perturbed run
    still agreed with ref in every discrete field), agree_upto = first iteration at which some perturbed run left the
    reference's discrete path (len(ref.trace) if none did)."""
    n = len(ref.trace)
    env = np.zeros(n)
    agree_upto = n
    for seed in seeds:
        res = run(UlpPerturbedBackend(seed))
        for i, (a, b) in enumerate(zip(res.trace, ref.trace)):
            if discrete(a) != discrete(b):
                agree_upto = min(agree_upto, i)
                break
            env[i] = max(env[i], abs(a["alpha"] - b["alpha"]) / max(1.0, abs(b["alpha"])))
        else:
            if len(res.trace) != n:
                agree_upto = min(agree_upto, min(len(res.trace), n))
    return env, agree_upto
