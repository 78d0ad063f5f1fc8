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
    config.addinivalue_line("markers", "heavy: large-model parity check (minutes); skipped once the session is past its time budget")


# The large-model parity checks (Recformer-large / BLaIR-large inference, the Recformer-large collaborative-merging step: 355-435 M
# parameters, several state dicts regenerated from seeds on the host) add about five minutes to the GPU suite (eight in all).  They run by
# default while the session is inside its time budget -- a slow box skips the remaining ones instead of running into a caller's time
# limit; MERGEREC_HEAVY_TESTS=1 always runs them, =0 never.  Their last outputs are kept in profiles/r02_recformer_realscale_parity.txt
# and profiles/r02_merge_train_step_parity.txt.
heavy = pytest.mark.heavy
_T0 = [None]
HEAVY_BUDGET_S = float(os.environ.get("MERGEREC_HEAVY_BUDGET_S", "420"))


def pytest_sessionstart(session):
    import time

    _T0[0] = time.monotonic()


def pytest_runtest_setup(item):
    if item.get_closest_marker("heavy") is None:
        return
    import time

    mode = os.environ.get("MERGEREC_HEAVY_TESTS", "")
    if mode == "0":
        pytest.skip("large-model parity check switched off (MERGEREC_HEAVY_TESTS=0); outputs of the last run are in profiles/")
    elapsed = time.monotonic() - (_T0[0] or time.monotonic())
    if mode != "1" and elapsed > HEAVY_BUDGET_S:
        pytest.skip(f"large-model parity check skipped: the session is {elapsed:.0f} s in, past its {HEAVY_BUDGET_S:.0f} s budget for starting one "
                    "(MERGEREC_HEAVY_TESTS=1 runs it regardless); outputs of the last run are in profiles/")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch

    return torch.load(GOLDEN / name, weights_only=False)
