"""tools/: load the DEVELOPMENT build of the library (make dev -> liboryon_hip_dev.so), the one that reads the ORYON_* variant switches.
Import this module before the first oryon_amd call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oryon_amd import _lib  # noqa: E402

_DEV = os.environ.get("ORYON_DEVLIB") or os.path.join(os.path.dirname(_lib.LIB_PATH), "liboryon_hip_dev.so")
if os.path.exists(_DEV):
    _lib.LIB_PATH = _DEV
else:
    print("tools/_devlib: liboryon_hip_dev.so not built (make -C oryon_amd/csrc dev): the shipped library ignores ORYON_* switches", file=sys.stderr)
