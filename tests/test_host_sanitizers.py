"""Host-side sanitizer build.  CPU only — this file is listed in .gpurunignore and never travels to a GPU
box: the GPU pool refuses anything that builds with sanitizers."""
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


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


