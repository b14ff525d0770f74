import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SESSION_T0 = __import__("time").time()          # tests/test_zz_gpu_default_routes.py budgets its optional steps against this


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")


def pytest_addoption(parser):
    parser.addoption("--all-cases", action="store_true", default=False,
                     help="also run the long emulated-device cases listed in tests/long_cases.txt (~100 CPU-minutes in all; use -n 8)")


def pytest_collection_modifyitems(config, items):
    """The default `pytest -m "not gpu"` run finishes in a few minutes: the cases listed in tests/long_cases.txt (the long
    parametrisations of the emulated-device tests; every test function keeps at least its cheapest case where that is cheap, chosen
    from measured durations by tools/select_long_cases.py) are skipped unless --all-cases / CSEG_TESTS_ALL=1 is given. GPU tests
    are never touched."""
    if config.getoption("--all-cases") or os.environ.get("CSEG_TESTS_ALL") == "1":
        return
    path = os.path.join(ROOT, "tests", "long_cases.txt")
    if not os.path.exists(path):
        return
    norm = lambda nid: nid[len("tests/"):] if nid.startswith("tests/") else nid          # (pytest started inside tests/)
    long_cases = {norm(line.split("\t")[0].strip()) for line in open(path) if line.strip() and not line.startswith("#")}
    skip = pytest.mark.skip(reason="long emulated-device case (tests/long_cases.txt): run with --all-cases or CSEG_TESTS_ALL=1")
    for item in items:
        if norm(item.nodeid) in long_cases and "gpu" not in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.hookimpl(trylast=True)
def pytest_terminal_summary(terminalreporter):
    """One compact line with what the last-sorted GPU file measured (tests/test_zz_gpu_default_routes.py): the driver keeps the
    tail of this output, which is the only way those first hardware results reach the next round."""
    mod = sys.modules.get("test_zz_gpu_default_routes")
    report = getattr(mod, "REPORT", None)
    if report:
        import json
        terminalreporter.write_line("CSEG_ZZ " + json.dumps(report, separators=(",", ":")))
