import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _have_gpu():
    try:
        from tsfresh_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _have_gpu():
        pytest.fail("this test is marked gpu but no HIP device / libtsfresh_amd.so is available "
                    "(the native path has no CPU fallback)")
    return True


def pytest_sessionfinish(session, exitstatus):
    """TSFA_PARITY_SKIPS_MD=<file>: per test, the cells tests/parity.py excluded, by calculator."""
    path = os.environ.get("TSFA_PARITY_SKIPS_MD")
    if not path:
        return
    try:
        import parity
    except Exception:
        return
    rows = []
    for test, e in sorted(parity.SKIP_LOG.items()):
        n = sum(e["skipped"].values())
        if e["cells"] == 0:
            continue
        by = ", ".join("%s %d" % (k, v) for k, v in sorted(e["skipped"].items(), key=lambda kv: -kv[1]))
        rows.append("| %s | %d | %d | %.3f %% | %s |" % (test.replace("tests/", ""), e["cells"], n, 100.0 * n / e["cells"], by or "-"))
    with open(path, "w") as f:
        f.write("cells compared through tests/parity.py::compare and cells its exclusion predicates skipped, per test\n\n")
        f.write("| test | cells | skipped | share | skipped by calculator |\n|---|---|---|---|---|\n")
        f.write("\n".join(rows) + "\n")
