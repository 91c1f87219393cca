"""CPU checks of the drop-in boundary: the library builds for gfx950, loads, and exports every
symbol include/enlsip_gn.h declares; without a GPU it fails loudly instead of falling back."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    import enlsip_gn._lib as L
    return L.load()


def test_header_symbols_exported(lib):
    import enlsip_gn._lib as L
    hdr = (ROOT / "include" / "enlsip_gn.h").read_text()
    declared = set(re.findall(r"\b(enlsip_gn_[a-z_A-Z0-9]+)\s*\(", hdr))
    declared -= {"enlsip_gn_context"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(L.PROTOTYPES), (declared ^ set(L.PROTOTYPES))
    assert lib.enlsip_gn_version() >= 100


def test_route_names_match_the_header(lib):
    """enlsip_gn_route_name (no GPU needed) speaks the header's ENLSIP_GN_ROUTE_* vocabulary, bit for bit — the vocabulary the
    dispatch-derived shape grid is written in (tests/dispatch_grid.py)."""
    import sys
    sys.path.insert(0, str(ROOT / "tests"))
    import dispatch_grid as dg
    from enlsip_gn.api import route_names
    assert route_names(lib) == dg.header_route_names()


def test_header_cites_reference_lines():
    hdr = (ROOT / "include" / "enlsip_gn.h").read_text()
    for cite in ("src/enlsip_functions.jl:206-234", "src/enlsip_functions.jl:116-153",
                 "src/enlsip_functions.jl:17-31", "src/enlsip_functions.jl:1249-1253"):
        assert cite in hdr


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from enlsip_gn import GNSolver, GNError
    with pytest.raises(GNError):
        GNSolver()


def test_product_does_not_import_oracle():
    for py in (ROOT / "enlsip.jl_amd").rglob("*.py"):
        txt = py.read_text()
        assert "oracle" not in txt.replace("CPU oracle under /oracle is test infrastructure only", ""), py
    for src in (ROOT / "enlsip.jl_amd" / "csrc").iterdir():
        assert "oracle/" not in src.read_text().replace("oracle/lapack_semantics.py", ""), src


def test_plain_c99_client_compiles_links_and_fails_loudly_without_gpu(lib, tmp_path):
    """tests/c_abi_client.c: a C99 translation unit (gcc -std=c99 -pedantic, no C++ / HIP headers) that includes the header,
    links the library and solves a small problem.  Here (no GPU) it must get a refusal WITH a message from enlsip_gn_create;
    on a GPU box the same binary is the `-m gpu` test below."""
    import subprocess, torch
    exe = tmp_path / "c_abi_client"
    libdir = ROOT / "enlsip.jl_amd" / "lib"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{ROOT / 'include'}",
                    str(ROOT / "tests" / "c_abi_client.c"), "-o", str(exe), f"-L{libdir}", "-lenlsip_gn", "-lm",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    if torch.cuda.is_available():
        assert out.returncode == 0, out.stdout
    else:
        assert out.returncode == 3 and "no usable HIP device" in out.stdout, out.stdout


@pytest.mark.gpu
def test_plain_c99_client_solves_on_gpu(lib, tmp_path):
    import subprocess
    exe = tmp_path / "c_abi_client"
    libdir = ROOT / "enlsip.jl_amd" / "lib"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{ROOT / 'include'}",
                    str(ROOT / "tests" / "c_abi_client.c"), "-o", str(exe), f"-L{libdir}", "-lenlsip_gn", "-lm",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "worst optimality residual" in out.stdout
