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
