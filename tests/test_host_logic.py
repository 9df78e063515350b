"""CPU tests of the product's host side (no GPU compute): the C-ABI library loads and exports every
symbol the headers declare, index math against the reference's golden rows and the oracle, the
synthetic-input generator against the oracle, descriptor helpers, loud failure without the library."""
import ctypes as C
import itertools
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dlaf():
    import dla_future_amd as d
    d.lib()
    return d


def declared_symbols():
    names = set()
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            if not f.endswith(".h"):
                continue
            txt = open(os.path.join(dirpath, f)).read()
            # MPI-typed entry points live in the optional MPI shim, not in libdlaf_mi355x.so
            txt = re.sub(r"#ifdef DLAF_MI355X_WITH_MPI.*?#endif", "", txt, flags=re.S)
            for m in re.finditer(r"DLAF_EXTERN_C\s+[\w\s\*]+?\b(\w+)\s*\(", txt):
                names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol(dlaf):
    L = C.CDLL(dlaf.lib_path())
    names = declared_symbols()
    assert {"dlaf_initialize", "dlaf_finalize", "dlaf_free_grid", "make_dlaf_descriptor", "dlaf_pdpotrf",
            "dlaf_pzpotrf", "dlaf_cholesky_factorization_d", "dlaf_cholesky_factorization_z"} <= names
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    # and the ctypes table of the Python mirror covers the same set
    from dla_future_amd.capi import SIGNATURES
    assert names == set(SIGNATURES), names ^ set(SIGNATURES)


def test_descriptor_struct_layout(dlaf):
    d = dlaf.make_descriptor(100, 32, 104, 1, 2)
    assert (d.m, d.n, d.mb, d.nb, d.isrc, d.jsrc, d.i, d.j, d.ld) == (100, 100, 32, 32, 1, 2, 0, 0, 104)
    assert C.sizeof(dlaf.DLAFDescriptor) == 9 * C.sizeof(C.c_int)
    desc9 = (C.c_int * 9)(1, 77, 100, 100, 32, 32, 1, 2, 104)
    r = dlaf.lib().make_dlaf_descriptor(100, 100, 1, 1, desc9)  # src/c_api/utils.cpp:25-33
    assert (r.m, r.n, r.mb, r.nb, r.isrc, r.jsrc, r.i, r.j, r.ld) == (100, 100, 32, 32, 1, 2, 0, 0, 104)


def test_distribution_golden_rows(dlaf, golden_dir):
    """test/unit/matrix/test_util_distribution.cpp:45-62, rows the Cholesky path can reach
    (tiles_per_block == 1, no offsets)."""
    from dla_future_amd import distribution as ds
    g = json.load(open(os.path.join(golden_dir, "distribution_rows.json")))
    used = 0
    for row in g["rows"]:
        p = dict(zip(g["columns"], row))
        if p["tiles_per_block"] != 1 or p["tile_offset"] or p["tile_element_offset"]:
            continue
        used += 1
        assert ds.rank_global_tile(p["global_tile"], p["grid_size"], p["src_rank"]) == p["rank_tile"]
        assert ds.local_tile_from_global_tile(p["global_tile"], p["grid_size"], p["rank"], p["src_rank"]) == p["local_tile"]
        assert ds.next_local_tile_from_global_tile(p["global_tile"], p["grid_size"], p["rank"],
                                                   p["src_rank"]) == p["local_tile_next"]
        if p["local_tile"] >= 0:
            assert ds.global_tile_from_local_tile(p["local_tile"], p["grid_size"], p["rank"], p["src_rank"]) == p["global_tile"]
    assert used == 10


def test_distribution_matches_oracle_exhaustively(dlaf, oracle):
    from dla_future_amd import distribution as ds
    for gs in (1, 2, 3, 4, 5):
        for src, rank in itertools.product(range(gs), range(gs)):
            for gt in range(0, 40):
                assert ds.rank_global_tile(gt, gs, src) == oracle.rank_global_tile(gt, gs, src)
                assert ds.local_tile_from_global_tile(gt, gs, rank, src) == oracle.local_tile_from_global_tile(gt, gs, rank, src)
                assert ds.next_local_tile_from_global_tile(gt, gs, rank, src) == \
                    oracle.next_local_tile_from_global_tile(gt, gs, rank, src)
            for n, nb in [(0, 2), (5, 8), (34, 13), (32, 5), (100, 7), (64, 8)]:
                assert ds.local_size(n, nb, gs, rank, src) == oracle.local_size(n, nb, gs, rank, src)
                assert ds.local_nr_tiles(n, nb, gs, rank, src) == oracle.local_nr_tiles(n, nb, gs, rank, src)


def test_generator_matches_oracle_single(dlaf, oracle):
    """set_random_hermitian_positive_definite of the product vs the oracle: real types bit for bit,
    complex within libm's sincos/sin+cos last-bit freedom."""
    grid = dlaf.Grid.single()
    for t, dt in oracle.DTYPES.items():
        for n, nb in [(34, 13), (100, 32), (16, 16)]:
            a = np.zeros((n, n), dtype=dt, order="F")
            dlaf.set_random_hermitian_positive_definite(grid, a, n, nb, nthreads=3)
            o = oracle.set_random_hpd(n, nb, dt)
            if t in "sd":
                assert np.array_equal(a, o)
            else:
                assert np.abs(a - o).max() <= 4 * oracle.eps_of(dt)
            assert np.array_equal(a, a.conj().T)
    grid.free()


def test_generator_local_parts_of_a_grid(dlaf, oracle):
    """Every rank of a (simulated) 2x3 grid generates exactly its block-cyclic share."""
    n, nb, pr, pc, sr, sc = 45, 8, 2, 3, 1, 2
    full = oracle.set_random_hpd(n, nb, np.float64)
    want = oracle.scatter(full, nb, pr, pc, sr, sc)
    keep = []
    for rank in range(pr * pc):
        def nobcast(axis, root, buf):
            raise AssertionError("no communication expected")
        g = dlaf.Grid.host(pr * pc, rank, pr, pc, "R", nobcast)
        keep.append(g)
        assert (g.myrow, g.mycol) == (rank // pc, rank % pc)
        rows, cols = g.local_shape(n, nb, sr, sc)
        loc = np.zeros((max(1, rows), max(1, cols)), order="F")[:rows, :cols]
        dlaf.set_random_hermitian_positive_definite(g, loc, n, nb, sr, sc)
        assert np.array_equal(loc, want[(g.myrow, g.mycol)])
    g = dlaf.Grid.host(6, 4, 2, 3, "C", lambda *a: None)
    assert (g.myrow, g.mycol) == (0, 2)  # column-major: rank = mycol*nprow + myrow


def test_contexts_count_down_from_int_max(dlaf):
    a = dlaf.Grid.single()
    b = dlaf.Grid.single()
    assert a.context <= 2 ** 31 - 1 and b.context != a.context and b.context > 2 ** 30  # src/c_api/grid.cpp:31
    a.free()
    b.free()


def test_a_live_grid_never_loses_its_context(dlaf):
    """create A, create B, free A, create C: C must not be handed B's context (the reference numbers contexts
    INT_MAX - #grids, src/c_api/grid.cpp:31, and never replaces a live entry); B keeps its shape, and freeing
    C leaves B alive."""
    lib = dlaf.lib()
    a = dlaf.Grid.single()
    b = dlaf.Grid.host(1, 0, 1, 1, "C", lambda *args: None)
    a.free()
    c = dlaf.Grid.single()
    assert c.context != b.context
    r = [C.c_int(-1) for _ in range(4)]
    assert lib.dlaf_mi355x_grid_info(b.context, *(C.byref(x) for x in r)) == 0
    assert [x.value for x in r] == [1, 1, 0, 0]
    freed = []
    cb = C.CFUNCTYPE(None, C.c_void_p)(lambda u: freed.append(u))
    assert lib.dlaf_mi355x_grid_on_free(c.context, C.cast(cb, C.c_void_p), 1234) == 0
    c.free()
    assert freed == [1234]
    assert lib.dlaf_mi355x_grid_info(b.context, *(C.byref(x) for x in r)) == 0
    assert lib.dlaf_mi355x_grid_info(c.context if c.context >= 0 else 5, *(C.byref(x) for x in r)) == -1
    b.free()


def _run(code):
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)


def test_unknown_context_terminates_like_upstream():
    """src/c_api/utils.cpp:55-68: an unknown context prints an error and terminates."""
    r = _run("import numpy as np, dla_future_amd as d\n"
             "from dla_future_amd.capi import lib\n"
             "a = np.eye(4, order='F')\n"
             "lib().dlaf_cholesky_factorization_d(12345, b'L', a.ctypes.data, d.make_descriptor(4, 2, 4))\n"
             "print('survived')")
    assert r.returncode != 0 and "survived" not in r.stdout
    assert "No DLA-Future grid for context 12345" in r.stderr


def test_bad_descriptor_terminates_like_upstream():
    """src/c_api/factorization/cholesky.h:37-38: i, j != 0 are asserted."""
    r = _run("import numpy as np, dla_future_amd as d\n"
             "from dla_future_amd.capi import lib\n"
             "g = d.Grid.single(); a = np.eye(4, order='F'); desc = d.make_descriptor(4, 2, 4); desc.i = 1\n"
             "lib().dlaf_cholesky_factorization_d(g.context, b'L', a.ctypes.data, desc)\n"
             "print('survived')")
    assert r.returncode != 0 and "survived" not in r.stdout and "sub-matrices are not supported" in r.stderr


@pytest.mark.parametrize("mutate,needle", [
    ("side = 'X'", "bad side/uplo/op/diag"),
    ("db.mb = 3", "B's blocks"),
    ("da.n = 5", "A must be square"),
    ("db.m = 7", "A is 6 x 6, B is 7 x 4"),
    ("db.isrc = 3", "outside the 1 x 1 grid"),
])
def test_triangular_solver_preconditions_terminate(mutate, needle):
    """include/dlaf/solver/triangular.h:43-57 / :93-107 assert their preconditions; so does this entry, before it
    touches the GPU (the checks run on a box without one)."""
    r = _run("import numpy as np, ctypes as C, dla_future_amd as d\n"
             "from dla_future_amd.capi import lib, DLAFDescriptor\n"
             "g = d.Grid.single(); a = np.eye(6, order='F'); b = np.ones((6, 4), order='F'); al = np.array([1.0])\n"
             "da = DLAFDescriptor(6, 6, 2, 2, 0, 0, 0, 0, 6); db = DLAFDescriptor(6, 4, 2, 2, 0, 0, 0, 0, 6); side = 'L'\n"
             f"{mutate}\n"
             "lib().dlaf_mi355x_triangular_solver_d(g.context, side.encode(), b'L', b'N', b'N', al.ctypes.data, "
             "a.ctypes.data, da, b.ctypes.data, db)\n"
             "print('survived')")
    assert r.returncode != 0 and "survived" not in r.stdout and needle in r.stderr, (r.stdout, r.stderr[-500:])


def test_missing_library_fails_loudly():
    r = _run("import dla_future_amd.capi as c\n"
             "c.lib_path = lambda: '/nonexistent/libdlaf_mi355x.so'\n"
             "try:\n    c.lib()\nexcept c.LibraryNotBuilt as e:\n    print('LOUD', e)\n")
    assert "LOUD" in r.stdout and "no CPU fallback" in r.stdout


def test_compute_without_gpu_fails_loudly():
    """On a box without a GPU the library must refuse to compute (no silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run("import numpy as np, dla_future_amd as d\n"
             "g = d.Grid.single(); a = np.eye(8, order='F')\n"
             "print('info', d.cholesky_factorization(g, 'L', a, 4))")
    assert r.returncode != 0 and "info" not in r.stdout
    assert "no HIP device" in r.stderr or "HIP error" in r.stderr


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dla_future_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py measures the HIP path only: on a box without a GPU it must say so and exit non-zero, never fall
    back to a CPU computation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout) and '"metric"' not in r.stdout
