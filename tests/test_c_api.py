"""The reference's C interface driven from plain C + MPI programs (tests/c_api/*.c), the way
test/unit/c_api/factorization/test_cholesky_c_api.cpp does through test_cholesky_c_api_wrapper.c.
Needs mpicc/mpiexec (MPICH ships with the image); skipped otherwise."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dla_future_amd", "lib")
MPICC = shutil.which("mpicc") or "/opt/conda/bin/mpicc"
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"


def build(name):
    if not (os.path.exists(MPICC) and os.path.exists(MPIEXEC)):
        pytest.skip("no MPI toolchain")
    shim = os.path.join(LIB, "libdlaf_mi355x_mpi.so")
    if os.path.exists(shim) and not os.path.exists(os.path.join(LIB, "mpi", "libmpi.so.12")):
        os.remove(shim)   # the private link directory did not travel with the snapshot: rebuild both
    if not os.path.exists(shim):
        subprocess.run(["make", "-C", os.path.join(ROOT, "dla_future_amd", "csrc"), "mpi"], check=True,
                       stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(LIB, "libdlaf_mi355x_mpi.so")):
        pytest.skip("libdlaf_mi355x_mpi.so not built")
    out = os.path.join(ROOT, "tests", "c_api", name)
    src = out + ".c"
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        # compiled as C (not C++), like the reference's wrapper, to prove the headers are C-clean; plain gcc
        # rather than the mpicc wrapper, whose -L<mpi>/lib would link against the MPI tree's old libstdc++
        inc = os.path.join(os.path.dirname(os.path.dirname(MPICC)), "include")
        subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", src, "-I", os.path.join(ROOT, "include"), "-I", inc,
                        "-DDLAF_MI355X_WITH_MPI", "-L", LIB, "-L", os.path.join(LIB, "mpi"), "-ldlaf_mi355x_mpi",
                        "-ldlaf_mi355x", "-lmpi", "-lm", f"-Wl,-rpath-link,{LIB}/mpi", f"-Wl,-rpath,{LIB}",
                        f"-Wl,-rpath,{LIB}/mpi", "-Wl,-rpath,/opt/rocm/lib", "-o", out], check=True)
    return out


BLACS_LIBS = ["-lmkl_blacs_intelmpi_lp64", "-lmkl_intel_lp64", "-lmkl_sequential", "-lmkl_core", "-ldl", "-lpthread"]


def build_blacs():
    """tests/c_api/test_blacs_grid.c against the image's MKL BLACS (MPICH ABI) -- the BLACS a ScaLAPACK caller has."""
    build("test_grid_mpi")  # (the shim)
    libdir = os.path.join(os.path.dirname(os.path.dirname(MPICC)), "lib")
    if not os.path.exists(os.path.join(libdir, "libmkl_blacs_intelmpi_lp64.so")):
        pytest.skip("no BLACS library in the image")
    out = os.path.join(ROOT, "tests", "c_api", "test_blacs_grid")
    src = out + ".c"
    # the MKL libraries are reached through links in the (untracked) lib/mpi directory, like libmpi itself: the
    # MPI tree's lib directory must stay out of the link and run paths (it carries an old libstdc++)
    for name in ("libmkl_blacs_intelmpi_lp64.so", "libmkl_intel_lp64.so", "libmkl_sequential.so", "libmkl_core.so",
                 "libmkl_intel_lp64.so.1", "libmkl_sequential.so.1", "libmkl_core.so.1", "libmkl_blacs_intelmpi_lp64.so.1"):
        link = os.path.join(LIB, "mpi", name)
        if os.path.exists(os.path.join(libdir, name)) and not os.path.lexists(link):
            os.symlink(os.path.join(libdir, name), link)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        inc = os.path.join(os.path.dirname(os.path.dirname(MPICC)), "include")
        subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", src, "-I", os.path.join(ROOT, "include"), "-I", inc,
                        "-DDLAF_MI355X_WITH_MPI", "-L", LIB, "-L", os.path.join(LIB, "mpi"),
                        "-ldlaf_mi355x_mpi", "-ldlaf_mi355x"] + BLACS_LIBS +
                       ["-lmpi", "-lm", f"-Wl,-rpath-link,{LIB}/mpi", f"-Wl,-rpath,{LIB}", f"-Wl,-rpath,{LIB}/mpi",
                        "-Wl,-rpath,/opt/rocm/lib", "-o", out], check=True)
    return out


def run(exe, nprow, npcol, order, timeout, env_extra=None, extra=None):
    env = dict(os.environ, DLAF_MI355X_MPI_TRANSPORT="host", DLAF_MI355X_DEVICE="0", OMP_NUM_THREADS="1")
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    r = subprocess.run([MPIEXEC, "-n", str(nprow * npcol), exe, str(nprow), str(npcol), order] + list(extra or []), cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=timeout)
    return r


@pytest.mark.parametrize("nprow,npcol,order", [(1, 1, "R"), (2, 2, "C"), (3, 2, "R"), (2, 3, "C")])
def test_mpi_grid_entry_points_cpu(nprow, npcol, order):
    exe = build("test_grid_mpi")
    r = run(exe, nprow, npcol, order, 120)
    assert r.returncode == 0 and "GRID_MPI_TEST OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.gpu
@pytest.mark.parametrize("nprow,npcol,order", [(1, 1, "R"), (1, 2, "R"), (2, 2, "C"),
                                               pytest.param(3, 2, "R", marks=pytest.mark.many_ranks)])
def test_pdpotrf_pzpotrf_from_c_with_mpi(nprow, npcol, order):
    from conftest import gpu_process_budget
    gpu_process_budget(nprow * npcol)
    exe = build("test_pdpotrf")
    r = run(exe, nprow, npcol, order, 600)
    assert r.returncode == 0 and "C_API_TEST OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("nprow,npcol,order", [(1, 1, "R"), (2, 2, "C")])
def test_pdsyevd_pzheevd_pdsygvd_from_c_with_mpi(nprow, npcol, order):
    """The eigensolver entries of the reference's C interface from a plain C caller under mpiexec
    (test/unit/c_api/eigensolver/test_eigensolver_c_api_wrapper.c:17-60): tests/c_api/test_pdsyevd.c."""
    from conftest import gpu_process_budget
    gpu_process_budget(nprow * npcol)
    exe = build("test_pdsyevd")
    r = run(exe, nprow, npcol, order, 600)
    assert r.returncode == 0 and "C_API_EIG_TEST OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_mpi_shim_exports_the_mpi_guarded_declarations():
    """Every prototype inside an #ifdef DLAF_MI355X_WITH_MPI block of include/ is exported by the shim."""
    import re
    build("test_grid_mpi")
    names = set()
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            txt = open(os.path.join(dirpath, f)).read()
            for blk in re.findall(r"#ifdef DLAF_MI355X_WITH_MPI(.*?)#endif", txt, flags=re.S):
                names |= {m.group(1) for m in re.finditer(r"DLAF_EXTERN_C\s+[\w\s\*]+?\b(\w+)\s*\(", blk)}
    assert {"dlaf_create_grid", "grid_ordering"} <= names
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIB, "libdlaf_mi355x_mpi.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert names <= exported, names - exported


@pytest.mark.parametrize("nprow,npcol,order", [(1, 1, "R"), (2, 2, "R"), (3, 2, "C"), (1, 2, "C")])
def test_grid_from_blacs_context_cpu(nprow, npcol, order):
    """dlaf_create_grid_from_blacs: the grid of an existing BLACS context, found again under that context."""
    exe = build_blacs()
    r = run(exe, nprow, npcol, order, 120)
    assert r.returncode == 0 and "BLACS_GRID_TEST OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.gpu
@pytest.mark.parametrize("nprow,npcol,order", [(1, 1, "R"), (2, 2, "C")])
def test_pdpotrf_on_a_blacs_context(nprow, npcol, order):
    """The ScaLAPACK application's call sequence: BLACS grid, dlaf_create_grid_from_blacs, dlaf_pdpotrf with the
    BLACS context in the descriptor."""
    exe = build_blacs()
    env = dict(os.environ, DLAF_MI355X_MPI_TRANSPORT="host", DLAF_MI355X_DEVICE="0", OMP_NUM_THREADS="1")
    env.pop("LOCAL_RANK", None)
    r = subprocess.run([MPIEXEC, "-n", str(nprow * npcol), exe, str(nprow), str(npcol), order, "factorize"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BLACS_GRID_TEST OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def build_plain():
    """tests/c_api/test_plain_link.c: NO macro of this repository, linked against -ldlaf_mi355x alone (+ the caller's MPI):
    the reference's headers declare dlaf_create_grid / grid_ordering unconditionally (grid.h:31,54) and a caller links
    -ldlaf; here <mpi.h> on the include path switches the declarations on and the core library forwards to the shim."""
    build("test_grid_mpi")  # makes sure the shim and its private link directory exist
    out = os.path.join(ROOT, "tests", "c_api", "test_plain_link")
    src = out + ".c"
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        inc = os.path.join(os.path.dirname(os.path.dirname(MPICC)), "include")
        subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", src, "-I", os.path.join(ROOT, "include"), "-I", inc, "-L", LIB,
                        "-L", os.path.join(LIB, "mpi"), "-ldlaf_mi355x", "-lmpi", "-lm", f"-Wl,-rpath-link,{LIB}/mpi",
                        f"-Wl,-rpath,{LIB}", f"-Wl,-rpath,{LIB}/mpi", "-Wl,-rpath,/opt/rocm/lib", "-o", out], check=True)
    needed = subprocess.run(["readelf", "-d", out], capture_output=True, text=True, check=True).stdout
    assert "libdlaf_mi355x.so" in needed and "libdlaf_mi355x_mpi.so" not in needed   # the shim is NOT linked
    return out


@pytest.mark.parametrize("nprow,npcol,order", [(1, 1, "R"), (2, 2, "C"), (3, 2, "R")])
def test_reference_call_sequence_links_against_the_core_library_alone_cpu(nprow, npcol, order):
    exe = build_plain()
    r = run(exe, nprow, npcol, order, 120, extra=["cpu"])
    assert r.returncode == 0 and "PLAIN_LINK_TEST OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("nprow,npcol,order", [(1, 1, "R"), (2, 2, "C")])
def test_reference_call_sequence_links_against_the_core_library_alone(nprow, npcol, order):
    from conftest import gpu_process_budget
    gpu_process_budget(nprow * npcol)
    exe = build_plain()
    r = run(exe, nprow, npcol, order, 600)
    assert r.returncode == 0 and "PLAIN_LINK_TEST OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
