"""The C++ host mirror (foo-dsp-bfir_amd/host/*.hpp: brutefir, fftw_convolver,
coeff::preprocess_coeff) used the way the reference's callers use the reference classes.
Compiles tests/cpp/test_host_mirror.cpp with g++ against libbfir_hip.so and runs it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")


def _compile(tmp_path, bfir):
    exe = str(tmp_path / "test_host_mirror")
    libdir = os.path.dirname(bfir.library_path())
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", SRC, "-o", exe, "-L" + libdir,
                    "-lbfir_hip", "-Wl,-rpath," + libdir], check=True)
    return exe


def test_host_mirror_compiles_without_hipcc(tmp_path, bfir):
    """The headers need only the C ABI: a plain g++ build must succeed (CPU check)."""
    assert os.path.exists(_compile(tmp_path, bfir))


@pytest.mark.gpu
def test_host_mirror_runs(tmp_path, bfir):
    exe = _compile(tmp_path, bfir)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(p.stdout[-4000:], p.stderr[-2000:])
    assert p.returncode == 0 and "ALL OK" in p.stdout
