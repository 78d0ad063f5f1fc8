import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "heavy: large-model parity check (minutes); runs first in the session, MERGEREC_HEAVY_TESTS=0 opts out")


# The large-model parity checks (Recformer-large / BLaIR-large inference, the Recformer-large collaborative-merging steps: 355-435 M
# parameters, state dicts regenerated from seeds on the host once per module) are DETERMINISTIC members of the GPU suite: they are moved
# to the front of the session, so which of them run no longer depends on how fast the box got through the other tests (a
# time-budget gate made "148 passed" unreproducible on a slower box).  MERGEREC_HEAVY_TESTS=0 is the one explicit opt-out (then every
# heavy test is reported as skipped, loudly, in the summary line).
heavy = pytest.mark.heavy


def pytest_collection_modifyitems(config, items):
    first = [it for it in items if it.get_closest_marker("heavy") is not None]
    rest = [it for it in items if it.get_closest_marker("heavy") is None]
    # keep module-scoped fixtures together: stable sort by file inside the heavy block
    first.sort(key=lambda it: str(it.fspath))
    items[:] = first + rest


def pytest_runtest_setup(item):
    if item.get_closest_marker("heavy") is not None and os.environ.get("MERGEREC_HEAVY_TESTS", "") == "0":
        pytest.skip("large-model parity check switched off by MERGEREC_HEAVY_TESTS=0")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    skipped = [r for r in terminalreporter.stats.get("skipped", []) if "MERGEREC_HEAVY_TESTS=0" in str(getattr(r, "longrepr", ""))]
    if skipped:
        terminalreporter.write_line(f"WARNING: {len(skipped)} large-model parity check(s) were switched off by MERGEREC_HEAVY_TESTS=0 -- "
                                    "this run does not cover BLaIR-large / Recformer-large", red=True, bold=True)


# ---- background regeneration of the seed-defined inputs of the real-dimension checks: tests/_prefetch.py (ONE module object, whichever
# name this file is imported under -- pytest loads it as `conftest`, the test modules as `tests.conftest`)
from tests._prefetch import prefetched, register_prefetch, seeded_state_dicts, start_prefetch, stop_prefetch  # noqa: E402,F401


def pytest_collection_finish(session):
    start_prefetch(session)


def pytest_sessionfinish(session, exitstatus):
    stop_prefetch(session)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch

    return torch.load(GOLDEN / name, weights_only=False)
