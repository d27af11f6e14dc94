"""oryon_amd: the MI355X-native Oryon match -> lift -> registration path."""
import os

# The step engine (csrc/engine.hip) drives four HIP streams of its own (gather, match, two registration streams) next to the
# caller's.  The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4): with five
# streams two of them share a queue, and a registration stream that shares the caller's queue runs its kernels behind the caller's
# event waits, i.e. the two registration streams serialise (measured: 5.0 instead of 4.7 ms per cfg2 step).  The variable is read
# when the runtime initialises (the first HIP call), so setting it at import time is early enough; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
