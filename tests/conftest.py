import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """The suite tests the library built from the sources next to it: `make` (a no-op when fresh) runs first, as in
    __graft_entry__.build(); when file times say "fresh" but the digest baked into the library differs from the
    sources (a snapshot that lost its mtimes), everything is rebuilt.  _lib.lib() raises on a mismatch anyway."""
    import shutil
    import subprocess
    if os.environ.get("D3F_LIB") or not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        return
    from denoising_diffusion_deep_fake_amd import _lib
    csrc = ROOT / "denoising_diffusion_deep_fake_amd" / "csrc"

    def stale():
        # the digest string baked into the file (no dlopen here: a library loaded now would stay mapped after a rebuild)
        return not _lib.LIB_PATH.exists() or _lib.source_digest().encode() not in _lib.LIB_PATH.read_bytes()

    if stale():
        subprocess.run(["make", "-C", str(csrc), "-j8"], check=True, stdout=subprocess.DEVNULL)
        if stale():
            subprocess.run(["make", "-B", "-C", str(csrc), "-j8"], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
