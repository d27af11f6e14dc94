"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


import oryon_amd  # noqa: E402
from oryon_amd import _lib as _oryon_lib  # noqa: E402

# The shipped library reads no environment variable.  Variant-vs-variant tests re-run themselves in a child interpreter against the
# DEVELOPMENT build (`make dev`: liboryon_hip_dev.so, the same sources with the ORYON_* kernel-variant switches compiled in); the child is
# told so with ORYON_TEST_DEV_LIB=1 - test infrastructure only, the product never looks at it.
DEV_LIB_PATH = os.path.join(os.path.dirname(_oryon_lib.LIB_PATH), "liboryon_hip_dev.so")
if os.environ.get("ORYON_TEST_DEV_LIB") == "1":
    assert os.path.exists(DEV_LIB_PATH), "liboryon_hip_dev.so missing: run `make -C oryon_amd/csrc dev` (or __graft_entry__.build())"
    _oryon_lib.LIB_PATH = DEV_LIB_PATH

oryon_amd.configure()        # hardware queues for the step engine's streams, before any test initialises HIP


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
