"""The C++ facade include/dlaf_mi355x/dlaf.hpp (the reference's C++ names over the C ABI): a C++17 program
patterned on test/unit/factorization/test_cholesky.cpp and test/unit/solver/test_triangular.cpp is compiled
with g++ here (CPU: the header is self-contained and links) and run on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dla_future_amd", "lib")
SRC = os.path.join(ROOT, "tests", "cpp_api", "test_cholesky_cpp.cpp")
EXE = os.path.join(ROOT, "tests", "cpp_api", "test_cholesky_cpp")


def build():
    newest = max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "dlaf_mi355x", "dlaf.hpp")))
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < newest:
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-L", LIB,
                        "-ldlaf_mi355x", f"-Wl,-rpath,{LIB}", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE], check=True)
    return EXE


def test_cpp_facade_compiles_and_links():
    assert os.path.exists(build())


@pytest.mark.gpu
def test_reference_style_cpp_tests_on_the_facade():
    r = subprocess.run([build()], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, DLAF_MI355X_DEVICE="0"))
    assert r.returncode == 0 and "CPP_API_TEST OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
