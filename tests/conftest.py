"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


import oryon_amd  # noqa: E402

oryon_amd.configure()        # hardware queues for the step engine's streams, before any test initialises HIP


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
