"""The device-side statement of SURVEY 8d's counter-based generator (enlsip_gn/workload.py, used by bench.py and the
full-size GPU tests) against the NumPy statement the oracle tests use (oracle/synth.py): same 64-bit streams, deviates equal to
an ulp or two of the transcendental functions."""
import numpy as np
import pytest

from oracle import synth

torch = pytest.importorskip("torch")


def test_uniform_bits_identical():
    from enlsip_gn import workload as wl
    x = np.arange(0, 5000, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15 % (1 << 63)) + np.uint64(12345)
    ref = synth.splitmix64(x)
    got = wl.splitmix64(torch.from_numpy(x.view(np.int64))).numpy().view(np.uint64)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("m,n,t", [(40, 7, 3), (256, 32, 4), (33, 5, 0)])
def test_batches_match_numpy_generator(m, n, t):
    from enlsip_gn import workload as wl
    J, rx, At, cx = wl.make_batch(11, 3, m, n, t, "cpu")
    for k in range(3):
        Jr, rr, Ar, cr = synth.make_problem(11 + k, m, n, t)
        assert np.abs(J[k].numpy().T - Jr).max() <= 1e-15 * 8
        assert np.abs(rx[k].numpy() - rr).max() <= 1e-15 * 8
        if t:
            assert np.abs(At[k].numpy() - Ar).max() <= 1e-15 * 8 and np.abs(cx[k].numpy() - cr).max() <= 1e-15 * 8


def test_tall_problem_in_column_pieces_and_row_blocks():
    from enlsip_gn import workload as wl
    m, n, t = 1000, 5, 2
    Jr, rr, Ar, cr = synth.make_problem(10, m, n, t)
    J, rx, At, cx = wl.make_batch(10, 1, m, n, t, "cpu", chunk_elems=2100)      # forces the column-piece path
    assert np.abs(J[0].numpy().T - Jr).max() <= 8e-15
    blocks = []
    for lo, hi in ((0, 300), (300, 650), (650, 1000)):
        Jl, rl, Al, cl = wl.make_row_block(10, m, n, t, lo, hi, "cpu", chunk_elems=800)
        assert np.abs(rl.numpy() - rr[lo:hi]).max() <= 8e-15 and np.abs(Al.numpy() - Ar).max() <= 8e-15
        blocks.append(Jl.numpy().T)
    assert np.abs(np.vstack(blocks) - Jr).max() <= 8e-15


def test_configs_match_baseline_json():
    import json
    from pathlib import Path
    from enlsip_gn import workload as wl
    cfgs = json.loads((Path(__file__).resolve().parents[1] / "BASELINE.json").read_text())["configs"]
    assert "n=512, m=4096, 64 eq" in cfgs[1] and (wl.CONFIGS["C2"]["m"], wl.CONFIGS["C2"]["n"], wl.CONFIGS["C2"]["t"]) == (4096, 512, 64)
    assert "1024 independent (n=64, m=512)" in cfgs[2] and wl.CONFIGS["C3"]["batch"] == 1024
    assert "n=1024, m=262144" in cfgs[3] and (wl.CONFIGS["C4"]["m"], wl.CONFIGS["C4"]["n"]) == (262144, 1024)
    assert "65536 (n=32, m=256)" in cfgs[4] and wl.CONFIGS["C5"]["batch"] * 8 == 65536
