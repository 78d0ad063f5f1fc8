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


# The large-model parity checks (Recformer-large / BLaIR-large inference, the Recformer-large collaborative-merging step: 355-435 M
# parameters, several state dicts regenerated from seeds on the host) add about five minutes to the GPU suite.  They run with
# MERGEREC_HEAVY_TESTS=1; their last outputs are kept in profiles/r02_recformer_realscale_parity.txt and profiles/r02_merge_train_step_parity.txt.
heavy = pytest.mark.skipif(os.environ.get("MERGEREC_HEAVY_TESTS", "0") != "1",
                           reason="large-model parity check (about five minutes in all): set MERGEREC_HEAVY_TESTS=1; outputs of the last run are in profiles/")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch

    return torch.load(GOLDEN / name, weights_only=False)
