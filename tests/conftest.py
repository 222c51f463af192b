"""Shared test plumbing: import paths, the `gpu` marker, oracle + library builders."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "src", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    so = ROOT / "oracle" / "libggc_oracle.so"
    srcs = list((ROOT / "oracle").glob("*.c")) + [ROOT / "oracle" / "ggc_oracle.h"]
    if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def gpu_ctx():
    """A libggc_hip.so context on cuda:0; fails loudly if the extension is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    from gcn_grabcut import _native
    return _native.get_context(0)
