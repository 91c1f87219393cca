"""Stratified shape list derived FROM the kernel-selection conditions of the library's host code.

`expected_route(batch, m, n, t)` restates, predicate by predicate, how enlsip_gn.hip and the launch helpers of the kernel headers
pick their kernels for a full-rank problem on a fresh handle (thresholds parsed from the C++ sources, not copied), in the
vocabulary of include/enlsip_gn.h's ENLSIP_GN_ROUTE_* bits.  `grid()` walks every boundary of those predicates — n2 + 1 <= 32 |
<= 64 | > 64; t = 0 | <= 16 | <= 63 | 64 | > 64; kp <= 32 | 64 | 128 | 256 | 448 | 512 | > 512; LDS area 8192 doubles; batch 1 |
< 128 | >= 128 — and returns cases whose expected routes together cover every bit of the header except those listed in
COVERED_ELSEWHERE (with the test that asserts them) — CPU test: tests/test_dispatch_grid.py::test_grid_covers_every_route_bit.
On the GPU every case is solved, its reported route (enlsip_gn_get_route) must CONTAIN the expected one — so this restatement
cannot drift from the library — and its results are compared with the oracle.  A new fast path gets a route bit in the header;
the CPU test then fails until a case reaches it.

Test infrastructure: nothing here is on the product path."""
from __future__ import annotations

import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "enlsip.jl_amd" / "csrc"


def _const(file: str, name: str) -> int:
    txt = (CSRC / file).read_text()
    mt = re.search(r"constexpr\s+(?:int|long long)\s+" + name + r"\s*=\s*([0-9]+)", txt)
    assert mt, (file, name)
    return int(mt.group(1))


CMAT = _const("gn_kernels_constraint.hpp", "CMAT_DOUBLES")       # LDS matrix area of the single-workgroup kernels (doubles)
KBLK = _const("gn_kernels_constraint.hpp", "KBLK")               # reflectors per compact-WY block of Q1
Q1R_MAXK = _const("gn_kernels_q1_rows.hpp", "Q1R_MAXK")          # most reflectors the lane-per-row J*Q1 kernels take
PB = _const("gn_kernels_caqr.hpp", "PB")                         # panel width
MAX_LAUNCH_BATCH = _const("enlsip_gn.hip", "GN_MAX_LAUNCH_BATCH")
PIPELINE_MIN = int(re.search(r"pipeline_min\s*=\s*([0-9]+)", (CSRC / "gn_context.hpp").read_text()).group(1))
PAIR_MIN_WGS = int(re.search(r"far_wgs >= ([0-9]+)", (CSRC / "enlsip_gn.hip").read_text()).group(1))


def header_route_names() -> list:
    """ENLSIP_GN_ROUTE_* of include/enlsip_gn.h in bit order, lower case without the prefix (= enlsip_gn_route_name)."""
    txt = (ROOT / "include" / "enlsip_gn.h").read_text()
    body = txt[txt.index("ENLSIP_GN_ROUTE_CONSTRAINT_WAVE32"):txt.index("ENLSIP_GN_ROUTE_COUNT")]
    return [nm.lower() for nm in re.findall(r"ENLSIP_GN_ROUTE_([A-Z0-9_]+)\s*(?:=\s*0)?,", body)]


# bits no grid case can reach on a fresh default handle: the test that asserts each of them instead
COVERED_ELSEWHERE = {
    "jq1_rows64": "leading dimensions beyond 2^23 only (one problem of 64 GB): not run",
    "jq1_plain": "tests/test_gpu_parity.py::test_mfma_and_reflector_updates_agree (ENLSIP_GN_UPDATE_REFLECTORS handle)",
    "sweep_reflectors": "tests/test_gpu_parity.py::test_mfma_and_reflector_updates_agree",
    "sweep_lookahead": "tests/test_gpu_full_configs.py::test_c4_full_size_eight_row_shards / test_lookahead_sweep_matches_the_one_stream_sweep",
    "sweep_upper_input": "tests/test_gpu_parity.py::test_tsqr_row_shards_match_single_solve (G = 1 combine)",
    "pivot_steps": "tests/test_gpu_robustness.py::test_hybrid_pivoted_qr_matches_the_launch_per_step_form (ENLSIP_GN_QRCP_HYBRID=0)",
    "chunked": "tests/test_gpu_full_configs.py::test_batch_above_launch_limit_small_shape",
    "rescaled": "tests/test_gpu_parity.py::test_extreme_magnitudes_match_lapack",
}


def _rows_bucket(prefix: str, rows: int) -> str:
    for lim, nm in ((32, "r1_256"), (64, "r1_512"), (128, "r2"), (256, "r4"), (512, "r8")):
        if rows <= lim:
            return f"{prefix}_{nm}"
    return f"{prefix}_r16"


def expected_route(batch: int, m: int, n: int, t: int, rank_deficient_A: bool = False, tile_rows: int = 512) -> set:
    """Route bits a fresh default handle must report for a batch of (m, n, t) problems whose A has full rank."""
    r = set()
    kA = min(n, t)
    n2 = n - kA
    kp = min(m, n2)
    # -- one launch set or several (solve_chunked / solve_launchable)
    small_uniform = n <= 64 and m <= 512
    split = batch >= PIPELINE_MIN and not small_uniform and batch <= MAX_LAUNCH_BATCH
    if batch > MAX_LAUNCH_BATCH:
        r.add("chunked")
    if split:
        r.add("pipeline_split")
    b_launch = (batch + 1) // 2 if split else batch          # problems per solve_dev
    # -- constraint stage (run_constraint_stage, launch_constraint, launch_constraint_small)
    fa_done = False
    if t > 64 and n * t > CMAT:
        r.add("constraint_dist")
        r.add(_rows_bucket("constraint_lds", max(n, t)))
    else:
        if 1 <= t <= 64 and n <= 512 and n * t > CMAT:
            r.add("constraint_reg4" if n <= 256 else "constraint_reg8")
            fa_done = True
        if n <= 64 and t <= 63 and not fa_done:
            r.add("constraint_wave32" if (n <= 32 and t <= 32) else "constraint_wave64")
        else:
            rows = max(t, 1) if fa_done else max(n, t)
            r.add(_rows_bucket("constraint_lds", rows))
            if not fa_done and n * t > CMAT:
                r.add("constraint_global")
    # -- J * Q1 (solve_dev)
    fused = m <= 256 and n <= 32 and kA <= Q1R_MAXK and 1 <= n2 < PB and m >= n2
    if fused:
        r.add("jq1_fused_small")
    elif n <= 64 and kA <= Q1R_MAXK:
        r.add("jq1_rows32" if n <= 32 else "jq1_rows2")
    elif n <= 512 and n % 128 == 0 and m % 32 == 0 and kA == KBLK and _vt_fits(m, n):
        r.add(f"jq1_v2_n{n}")
    else:
        r.add("jq1_mfma")
    # -- CAQR sweep (make_plan, run_caqr)
    npan = (kp + PB - 1) // PB
    if npan > 0 and not fused:
        rpl_rows = 256 if m <= 256 else tile_rows
        r.add("sweep_tile256" if rpl_rows == 256 else "sweep_tile512")
        tiles = (max(m, 1) + rpl_rows - 1) // rpl_rows
        npan_max = (min(m, n) + PB - 1) // PB
        pair = npan_max >= 3 and b_launch * tiles * ((max(n2, 1) + 31) // 32) >= PAIR_MIN_WGS
        last_bw = kp - (npan - 1) * PB
        passenger = last_bw < PB and kp == n2              # the last, narrow panel: only d to its right
        if pair:
            if npan >= 2:
                r.add("sweep_pairs")
            if npan % 2 == 1:
                r.add("sweep_passenger" if passenger else "sweep_plain")
        else:
            if npan > 1 or not passenger:
                r.add("sweep_plain")
            if passenger:
                r.add("sweep_passenger")
        if tiles > 1:
            r.add("sweep_tree")
    # -- pivoted QR of R0 + solves (solve_dev, launch_pivot_small, launch_pivot, run_qrcp_block)
    if kp * (n2 + 1) > CMAT:
        if kp > 512:
            r.add("pivot_hybrid")
            kp_blk = 512 - ((kp - 512) & 1)
        else:
            r.add("pivot_blocks")
            kp_blk = kp
        if kp_blk > 256:
            r.add("pivot_blocks_448" if kp_blk <= 448 else "pivot_blocks_512")
        if kp_blk > 128:
            r.add("pivot_blocks_256")
        r.add("pivot_blocks_128")
        r.add(_rows_bucket("pivot_lds", min(m, n)))
    elif kp <= 64 and n2 + 1 <= 64:
        if kp <= 32 and n2 + 1 <= 32 and b_launch > 1:
            r.add("pivot_wave2")
        else:
            r.add("pivot_wave32" if kp <= 32 else "pivot_wave64")
    else:
        r.add(_rows_bucket("pivot_lds", min(m, n)))
    if rank_deficient_A:
        r.add("second_attempt")
    return r


def _vt_fits(m: int, n: int) -> bool:
    """launch_jq1_v2 parks V T' in the working matrix of the pivoted QR (make_plan: sM); tiny m leaves no room."""
    ldr = (max(min(m, n), 1) + 7) // 8 * 8
    return ldr * (n + 1 + 33) >= n * KBLK


def grid() -> list:
    """Cases (dicts: batch, m, n, t, kind): every value of every dispatch axis at least once, most pairs of neighbouring axes.
    Shapes stay small (the oracle runs on every case; batches compare a sample of their problems)."""
    cases = []

    def add(batch, m, n, t, kind="full"):
        cases.append(dict(batch=batch, m=m, n=n, t=t, kind=kind))

    # constraint stage x J*Q1 kernels: t classes 0 | <= 16 | <= 63 | 64 | > 64 against n classes <= 32 | <= 64 | <= 256 | <= 512 | > 512
    for n in (24, 32, 33, 64, 65, 128, 200, 256, 257, 384, 512, 513, 700):
        for t in (0, 1, Q1R_MAXK, Q1R_MAXK + 1, 63, 64, 65, 150):
            if t > n + 8:
                continue
            m = n + 37 + (n % 5)
            if n % 128 == 0 and t == 64:
                m = 32 * ((n + 64) // 32)                     # the fast J*Q1 path wants m a multiple of 32
            add(1, m, n, t)
    # one-tile / fused / passenger shapes and the wave kernels' limits (n2 + 1 <= 32 | <= 64 | > 64; kp <= 32 | <= 64 | > 64)
    for (m, n, t) in ((256, 32, 4), (200, 30, 7), (256, 32, 0), (257, 32, 4), (40, 32, 4), (20, 32, 4), (300, 35, 4), (300, 63, 0),
                      (300, 64, 1), (300, 66, 1), (33, 80, 10), (64, 80, 10), (65, 80, 10), (90, 120, 50), (512, 64, 8), (513, 64, 8)):
        add(1, m, n, t)
        add(5, m, n, t)
    # pivoted QR of R0: LDS form up to 8192 doubles, register blocks by row count, hybrid beyond 512 rows
    for (m, n, t) in ((600, 90, 0), (600, 91, 0), (700, 128, 0), (700, 129, 3), (900, 256, 0), (900, 257, 1), (1000, 448, 0),
                      (1000, 449, 1), (1100, 512, 0), (1100, 513, 0), (700, 600, 20), (300, 500, 40), (100, 300, 12), (20, 100, 4), (50, 150, 0)):
        add(1, m, n, t)
    # batches: pipeline split from 128 problems (not for the small uniform shapes), panel pairs from 8192 far workgroups
    add(127, 300, 66, 5)
    add(128, 300, 66, 5)
    add(130, 512, 64, 8)                                      # small uniform: never split
    add(128, 256, 32, 4)
    add(140, 700, 80, 70)
    add(130, 4096, 576, 64)                                   # 65 problems x 8 tiles x 16 column blocks >= 8192 per half: pairs
    add(64, 4096, 512, 0)                                     # 64 x 8 x 16 = 8192: pairs without the split
    add(96, 2048, 256, 32)                                    # below it: plain sweep
    # kinds that change the route or the arithmetic: rank-deficient A (second attempt), rank-deficient / graded J
    for (m, n, t) in ((300, 40, 6), (700, 130, 20), (256, 32, 5), (900, 300, 40)):
        add(1, m, n, t, "rankdefA")
        add(1, m, n, t, "rankdefJ")
        add(1, m, n, t, "graded")
    add(6, 300, 40, 6, "rankdefA")
    add(1, 20, 50, 4, "graded")                               # wide graded J (m < n2): skipped by the generator until round 5
    add(1, 40, 200, 30, "graded")
    return cases
