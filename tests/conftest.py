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


# The oracle / golden PARITY files run first, in this order; property and debug-aid files (determinism, sanitizer, trainer plumbing) after
# them.  Round 5's driver run stopped (-x) at one flaky property test that sorts alphabetically before every file that pins SURVEY.md 8(a):
# parity evidence must not depend on a debug aid.  (The flaky test's cause was found and fixed — tests/test_determinism.py — this is belt
# and braces, not the fix.)
_PARITY_FIRST = ["test_oracle", "test_abi", "test_gpu_kernels", "test_ss2d_fused", "test_ss2d_deep", "test_ss2d_glue", "test_mlp", "test_modules",
                 "test_fullsize", "test_loss", "test_metric", "test_convgemm", "test_mpd", "test_trainstep", "test_ckpt_fixture", "test_tester"]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_PARITY_FIRST)}
    items.sort(key=lambda it: rank.get(os.path.splitext(os.path.basename(str(it.fspath)))[0], len(rank)))     # (stable: order inside a file kept)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("VMASR_PARITY_TABLE")
    if path:
        import errtable
        errtable.write(path)
