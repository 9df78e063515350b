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


MINI_SRC = os.path.join(ROOT, "miniapp", "miniapp_cholesky.cpp")
MINI_EXE = os.path.join(ROOT, "miniapp", "miniapp_cholesky")
MINI_MPI = os.path.join(ROOT, "miniapp", "miniapp_cholesky_mpi")


def build_miniapp(mpi=False, name="miniapp_cholesky"):
    src = os.path.join(ROOT, "miniapp", name + ".cpp")
    exe = os.path.join(ROOT, "miniapp", name + ("_mpi" if mpi else ""))
    return _build_miniapp(src, exe, mpi)


def _build_miniapp(MINI_SRC, exe, mpi):
    newest = max(os.path.getmtime(MINI_SRC), os.path.getmtime(os.path.join(ROOT, "include", "dlaf_mi355x", "dlaf.hpp")))
    if os.path.exists(exe) and os.path.getmtime(exe) >= newest:
        return exe
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), MINI_SRC, "-L", LIB]
    if mpi:
        import test_c_api
        test_c_api.build("test_grid_mpi")   # makes sure the MPI shim and its private link directory exist
        inc = os.path.join(os.path.dirname(os.path.dirname(test_c_api.MPICC)), "include")
        cmd += ["-DDLAF_MI355X_WITH_MPI", "-I", inc, "-L", os.path.join(LIB, "mpi"), "-ldlaf_mi355x_mpi", "-lmpi",
                f"-Wl,-rpath-link,{LIB}/mpi", f"-Wl,-rpath,{LIB}/mpi"]
    cmd += ["-ldlaf_mi355x", f"-Wl,-rpath,{LIB}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def test_miniapp_compiles():
    assert os.path.exists(build_miniapp())
    assert os.path.exists(build_miniapp(name="miniapp_triangular_solver"))
    assert os.path.exists(build_miniapp(name="miniapp_gen_to_std"))
    assert os.path.exists(build_miniapp(name="miniapp_reduction_to_band"))
    assert os.path.exists(build_miniapp(name="miniapp_eigensolver"))


def check_miniapp_output(out, nruns, nchecks):
    import re
    runs = re.findall(r"^\[(\d+)\] ([0-9.e+-]+)s ([0-9.e+-]+)GFlop/s (\w+) \((\d+), (\d+)\) \((\d+), (\d+)\) \((\d+), (\d+)\) 1 GPU$",
                      out, flags=re.M)
    assert len(runs) == nruns, out
    assert all(float(r[2]) > 0 for r in runs)
    assert out.count("CSVData-2, run, ") == nruns
    checks = re.findall(r"^(ERROR: |Warning: )?Max Diff / Max A: ([0-9.e+-]+)$", out, flags=re.M)
    assert len(checks) == nchecks and all(c[0] == "" for c in checks), out


@pytest.mark.gpu
def test_miniapp_cholesky_reference_cli_and_output():
    exe = build_miniapp()
    for t, uplo in (("d", "L"), ("z", "U")):
        r = subprocess.run([exe, "--matrix-size", "2048", "--block-size=256", "--nruns", "2", "--nwarmups", "1", "--type", t,
                            "--uplo", uplo, "--check-result", "all", "--csv"], cwd=ROOT, capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, DLAF_MI355X_DEVICE="0"))
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        check_miniapp_output(r.stdout, 2, 3)   # "all" also checks the warm-up run, as upstream does


@pytest.mark.gpu
def test_miniapp_cholesky_mpi_grid():
    import test_c_api
    exe = build_miniapp(mpi=True)
    env = dict(os.environ, DLAF_MI355X_MPI_TRANSPORT="host", DLAF_MI355X_DEVICE="0", OMP_NUM_THREADS="1")
    env.pop("LOCAL_RANK", None)
    r = subprocess.run([test_c_api.MPIEXEC, "-n", "4", exe, "--matrix-size", "1500", "--block-size", "128", "--grid-rows", "2",
                        "--grid-cols", "2", "--nruns", "2", "--check-result", "last", "--csv"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("GFlop/s d") == 2 and "(2, 2)" in r.stdout and "ERROR" not in r.stdout and \
        r.stdout.count("Max Diff / Max A") == 1, r.stdout


@pytest.mark.gpu
def test_miniapp_triangular_solver():
    import re
    exe = build_miniapp(name="miniapp_triangular_solver")
    r = subprocess.run([exe, "--m", "1500", "--n", "700", "--mb", "128", "--nb", "128", "--side", "L", "--uplo", "L", "--op", "N",
                        "--nruns", "2"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, DLAF_MI355X_DEVICE="0"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert len(re.findall(r"^\[\d+\] [0-9.e+-]+s [0-9.e+-]+GFlop/s dLLNN \(1500, 700\) \(128, 128\) \(1, 1\) 1 GPU", r.stdout,
                          flags=re.M)) == 2, r.stdout
    resid = float(re.search(r"corner: ([0-9.e+-]+)", r.stdout).group(1))
    assert resid < 1e-11, r.stdout


@pytest.mark.gpu
def test_miniapp_gen_to_std():
    """miniapp_gen_to_std.cpp: the reference's options and result lines; both operands resident on the device."""
    import re
    exe = build_miniapp(name="miniapp_gen_to_std")
    for tp, uplo in (("d", "L"), ("z", "U")):
        r = subprocess.run([exe, "--matrix-size", "1500", "--block-size", "128", "--type", tp, "--uplo", uplo, "--nruns", "2",
                            "--csv"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, DLAF_MI355X_DEVICE="0"))
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        assert len(re.findall(r"^\[\d+\] [0-9.e+-]+s [0-9.e+-]+GFlop/s %s%s \(1500, 1500\) \(128, 128\) \(1, 1\) 1 GPU" % (tp, uplo),
                              r.stdout, flags=re.M)) == 2, r.stdout
        assert r.stdout.count("CSVData-2, run, ") == 2, r.stdout


@pytest.mark.gpu
def test_miniapp_reduction_to_band():
    """miniapp_reduction_to_band.cpp: the reference's options (incl. --band-size) and result lines; the matrix resident
    on the device in the timed window."""
    import re
    exe = build_miniapp(name="miniapp_reduction_to_band")
    for tp, band in (("d", "64"), ("z", "128")):
        r = subprocess.run([exe, "--matrix-size", "1500", "--block-size", "128", "--band-size", band, "--type", tp, "--nruns", "2",
                            "--csv"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, DLAF_MI355X_DEVICE="0"))
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        assert len(re.findall(r"^\[\d+\] [0-9.e+-]+s [0-9.e+-]+GFlop/s %s \(1500, 1500\) \(128, 128\) %s \(1, 1\) 1 GPU" % (tp, band),
                              r.stdout, flags=re.M)) == 2, r.stdout
        assert r.stdout.count("CSVData-2, run, ") == 2 and "band_size, " + band in r.stdout, r.stdout


@pytest.mark.gpu
def test_miniapp_eigensolver():
    """miniapp_eigensolver.cpp: the reference's options and result lines, its checker (--check-result) on one process."""
    import re
    exe = build_miniapp(name="miniapp_eigensolver")
    for tp in ("d", "z"):
        r = subprocess.run([exe, "--matrix-size", "700", "--block-size", "128", "--type", tp, "--nruns", "2", "--check-result", "last",
                            "--csv"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, DLAF_MI355X_DEVICE="0"))
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        assert len(re.findall(r"^\[\d+\] [0-9.e+-]+s %s \(700, 700\) \(128, 128\) \(1, 1\) 1 GPU" % tp, r.stdout, flags=re.M)) == 2, r.stdout
        assert r.stdout.count("Check: OK") == 1 and "Check: ERROR" not in r.stdout, r.stdout
        assert r.stdout.count("CSVData-2, run, ") == 2, r.stdout
