import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/libhs_oracle.so) -- the checker, never the thing under test."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


_poisoned_once = []


def _poison_free_hbm(pattern):
    """Fill free HBM with a byte and give it back: what the library allocates next comes back holding that byte
    instead of the zeros of fresh pages -- reads of memory nobody wrote show up as wrong results (there is no GPU
    sanitizer on this pool).  Nearly all of it before the session's first GPU test, HS_TEST_POISON_GIB (default 8)
    before every other one: an allocation of 280 GB costs seconds."""
    import torch
    if not torch.cuda.is_available():
        return
    free, _ = torch.cuda.mem_get_info()
    want = int(free * 0.95)
    if _poisoned_once:
        want = min(want, int(float(os.environ.get("HS_TEST_POISON_GIB", "8")) * (1 << 30)))
    _poisoned_once.append(1)
    left, chunks = want, []
    while left > (1 << 30):
        size = min(left, 16 << 30)
        try:
            chunks.append(torch.empty(size, dtype=torch.uint8, device="cuda").fill_(pattern))
        except Exception:
            break
        left -= size
    torch.cuda.synchronize()
    del chunks
    torch.cuda.empty_cache()


@pytest.fixture(autouse=True)
def _poisoned_memory(request):
    """HS_TEST_POISON=<byte>: every GPU test starts on HBM filled with that byte (see _poison_free_hbm)."""
    pat = os.environ.get("HS_TEST_POISON")
    if pat is not None and request.node.get_closest_marker("gpu") is not None:
        _poison_free_hbm(int(pat, 0) & 0xff)
    yield
