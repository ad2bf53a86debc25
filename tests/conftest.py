import os
import sys
import zlib

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Collection order under `-x`: the hot-path parity files first (SURVEY.md 8a rows a4-a8 against the oracle), then the callers
# either side of the path, peripheral rows (8f) last -- a failure in a nicety must not hide the hot path.
ORDER = ["test_gpu_oracle_fullsize", "test_gpu_parity", "test_gpu_decode", "test_gpu_midm", "test_gpu_gemm_v6", "test_gpu_cdna4",
         "test_gpu_fullsize", "test_gpu_splitk", "test_gpu_fused_mlp", "test_w3", "test_moe",
         "test_repacker", "test_loader", "test_engine_cache", "test_gpu_multidevice", "test_oneshot", "test_fused_norm"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return ORDER.index(name) if name in ORDER else len(ORDER)
    items.sort(key=rank)  # stable: keeps the order inside a file
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_per_test(request):
    """global torch generators (CPU and device) seeded from the test's own id: data never depend on which tests ran before."""
    seed = zlib.crc32(request.node.nodeid.encode()) + 1000003 * int(os.environ.get("AWQ_TEST_SEED", "0"))
    torch.manual_seed(seed)
    yield


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def as_t(arr, dtype):
    """npz int16 bit pattern -> torch tensor of dtype (fp16/bf16), or plain tensor."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    return t.view(dtype) if dtype in (torch.float16, torch.bfloat16) else t
