import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fresh_parent: starts one-process copies on the GPU; runs before this process does GPU work of its own")
    config.addinivalue_line("markers", "many_ranks: starts >= 6 processes on the one GPU; runs before every "
                                       "in-process GPU test (the box allows 6 processes on the card)")


def pytest_collection_modifyitems(config, items):
    """The 6-rank grids of the reference's test suite (grids_6_ranks.h) put 6 worker processes on the one
    GPU, which is the box's limit: they must run while this pytest process has not opened the device yet.
    Whatever the selection (-k, file order), they are moved in front of every other test; when they cannot
    run they FAIL (gpu_process_budget), they are never skipped."""
    first = [it for it in items if it.get_closest_marker("many_ranks")]
    # fresh_parent: tests that put several one-process copies on the GPU and are ten times slower once this process has
    # run GPU work of its own (26 s alone, 230 - 280 s behind the eigensolver tests; not the workspace pool: releasing it
    # changed nothing): right behind the six-rank grids
    second = [it for it in items if it.get_closest_marker("fresh_parent") and not it.get_closest_marker("many_ranks")]
    if first or second:
        rest = [it for it in items if not it.get_closest_marker("many_ranks") and not it.get_closest_marker("fresh_parent")]
        items[:] = first + second + rest


def gpu_open_in_this_process():
    capi = sys.modules.get("dla_future_amd.capi")
    torch = sys.modules.get("torch")
    return bool((capi is not None and getattr(capi, "_lib", None) is not None) or
                (torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized()))


def gpu_process_budget(nprocs):
    """Fail (not skip) when `nprocs` more GPU processes would exceed the box's limit of 6 on the card."""
    if nprocs + (1 if gpu_open_in_this_process() else 0) > 6:
        pytest.fail(f"{nprocs} ranks + this pytest process (which already opened the GPU) exceed the limit of 6 "
                    "processes on the card; conftest orders many_ranks tests first -- an earlier test must have "
                    "opened the device outside that order")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
