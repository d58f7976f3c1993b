import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


class Margins:
    """Parity checks that record how much room they have: check(label, achieved, bound) asserts achieved <= bound and appends the pair to
    PDP_MARGINS_FILE (default gpurun_out/parity_margins.txt under the repository root when that directory can be created) - the file
    profiles/r03_parity_margins.txt is a copy of one GPU-box run.  Bounds are BASELINE.md section 3's stated tolerances, or ten times the
    error a row achieved when that is tighter."""

    def __init__(self):
        path = os.environ.get("PDP_MARGINS_FILE")
        if path is None:
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                path = os.path.join(ROOT, "gpurun_out", "parity_margins.txt")
            except OSError:
                path = ""
        self.path = path

    def check(self, label, achieved, bound):
        achieved, bound = float(achieved), float(bound)
        if self.path:
            try:
                with open(self.path, "a") as f:
                    f.write("%-110s achieved %.3e   bound %.1e   margin %s\n" % (label, achieved, bound, "x%.1f" % (bound / achieved) if achieved > 0 else "(exact)"))
            except OSError:
                pass
        assert achieved <= bound, "%s: %.3e > %.1e" % (label, achieved, bound)


@pytest.fixture(scope="session")
def margins():
    return Margins()
