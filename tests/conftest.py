import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _install_abort_trace():
    """tests/native/abort_trace.c: the native backtrace of the thread that calls abort() (or faults), written to a dup of the
    original stderr. Installed once, before any test; a box without gcc just goes without it."""
    import ctypes
    src = ROOT / "tests" / "native" / "abort_trace.c"
    out = ROOT / "tests" / "native" / "libabort_trace.so"
    try:
        if not out.exists() or out.stat().st_mtime < src.stat().st_mtime:
            subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-shared", "-o", str(out), str(src)], check=True)
        lib = ctypes.CDLL(str(out))
        if lib.abort_trace_install(os.dup(2)) != 0:
            print("[conftest] abort_trace_install failed", file=sys.stderr)
    except Exception as e:
        print(f"[conftest] no native abort trace: {e}", file=sys.stderr)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _install_abort_trace()


@pytest.fixture(scope="session", autouse=True)
def _built_cpu_libs():
    """CPU-side artefacts (oracle + host scene library) are built on demand; the HIP library is
    built by __graft_entry__.build() and is only needed by -m gpu tests."""
    import __graft_entry__ as ge
    ge.build_cpu()
    # The HIP library and the CLI normally arrive pre-built (they travel with the tree); rebuild them if a
    # checkout without build artefacts is being tested and hipcc is there. Never substitute anything else.
    # Both calls are no-ops when the binaries are newer than every source they are built from, so a kernel edit can
    # never be tested against a stale library.
    try:
        ge.build_device()
        ge.build_cli()
    except Exception as e:   # CPU-only tests still run; GPU/ABI tests will fail loudly on the missing library
        print(f"[conftest] could not build the device library: {e}")
