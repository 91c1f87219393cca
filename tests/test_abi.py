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


def test_host_code_under_address_and_undefined_sanitizers(tmp_path):
    """SURVEY section 5 (sanitizers): the HOST side of the library (plan / workspace carve, argument checks, error strings) built
    with -fsanitize=address,undefined (device code unchanged: -fno-gpu-sanitize) and driven by the plain-C client.  Without a GPU
    the client walks the refusal path of enlsip_gn_create and the NULL-handle paths; the sanitizers must stay silent (no leak,
    no invalid access, no undefined behaviour) and the exit code must be the client's own."""
    import os, shutil, subprocess, torch
    if torch.cuda.is_available():
        pytest.skip("sanitizer builds run on the CPU only (the GPU pool refuses sanitizer runs)")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not (os.path.exists(clang) and shutil.which("hipcc")):
        pytest.skip("ROCm clang / hipcc not found")
    so = tmp_path / "libenlsip_gn_asan.so"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fsanitize=address,undefined",
                    "-fno-gpu-sanitize", f"-I{ROOT / 'include'}", "-o", str(so), str(ROOT / "enlsip.jl_amd" / "csrc" / "enlsip_gn.hip"),
                    "-ldl"], check=True, capture_output=True, timeout=900)
    exe = tmp_path / "client_asan"
    subprocess.run([clang, "-std=c99", "-g", "-fsanitize=address,undefined", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c_abi_client.c"),
                    "-o", str(exe), f"-L{tmp_path}", "-lenlsip_gn_asan", "-lm", f"-Wl,-rpath,{tmp_path}", "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, capture_output=True, timeout=300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 3 and "no usable HIP device" in out.stdout, out.stdout + out.stderr
    assert "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr and "LeakSanitizer" not in out.stderr, out.stderr


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
