import os

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # vm_asr_amd/hip_env.py: before the GPU is initialised
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")   # vm_asr_amd/hip_env.py: stream-K GEMMs of two streams can stall the device
# the fp32 parity suite adjudicates forward AND backward with the float64-accumulating Linear (vm_asr_amd/linear.py:_use_f64acc; the
# product's default keeps it to gradient-free evaluation, where the LSD parity claim lives)
os.environ.setdefault("VMASR_LINEAR_F64ACC", "1")
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
